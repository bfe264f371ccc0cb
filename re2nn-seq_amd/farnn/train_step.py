"""The training step of the decomposed i-FST on the HIP path, as a torch.autograd.Function around
farnn_decomp_ifst_train_step (include/farnn.h).

What the reference does in FARNN_S_D_W_I_S.forward_local(train=True) + loss.backward()
(model_decompose_single.py:207-304, train_decompose.py:186-190) is split like this: the word table
Vgen = V_embed*beta + act(E G)*(1-beta) (model_decompose.py:222-241) is built for the whole vocabulary
by ordinary torch ops (one [V,D]x[D,R] product: its gradient flows to the embedding, the bridge matrix,
V_embed and beta through torch's autograd); everything that depends on the batch -- both chains, the
scores, the cross-entropy and the back-propagation through time -- is one library call.  The library
computes loss and all gradients in its forward call (the stash lives in its workspace); backward()
only hands them out, scaled by the incoming gradient.

Scope: farnn = 0/1/2, sum semiring, CE1 loss or (use_crf) the CRF negative log-likelihood.
"""
import torch

from .. import _lib


GATE_NAMES = ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')


class _DecompIfstTrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tc, ntok, x, lengths, labels, P, Vgen, S1, S2, W, Cmat, h0, hT, trans, *gates):
        dev = Vgen.device
        if dev.type != 'cuda':
            raise _lib.FarnnError('the training step runs on the HIP device only (no CPU fallback)')
        ws = [t.detach().contiguous().float() for t in (Vgen, S1, S2, W, Cmat, h0, hT)]
        Pc = None if P is None else P.detach().contiguous().float()
        tr = None if trans is None else trans.detach().contiguous().float()
        gs = [g.detach().contiguous().float() for g in gates]
        B, L = x.shape
        if ntok is None:            # counted on the host when the lengths live there (no device round trip in the step)
            ntok = int(lengths.clamp(0, L).sum())
        x = x.to(dev).contiguous()
        lengths = lengths.to(dev).contiguous()
        labels = labels.to(dev).contiguous()
        if ntok <= 0:
            raise ValueError('empty batch')
        grads = [torch.empty_like(t) for t in ws]
        gtr = None if tr is None else torch.empty_like(tr)
        ggs = [torch.empty_like(g) for g in gs]
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        tags = torch.empty((B, L), dtype=torch.int32, device=dev)
        names = ('Vgen', 'S1', 'S2', 'W', 'C', 'h0', 'hT')
        weights = {n: t.data_ptr() for n, t in zip(names, ws)}
        weights['P'] = None if Pc is None else Pc.data_ptr()
        weights['crf_trans'] = None if tr is None else tr.data_ptr()
        outputs = {'d' + n: g.data_ptr() for n, g in zip(names, grads)}
        outputs['loss'] = loss.data_ptr()
        outputs['tags'] = tags.data_ptr()
        outputs['dtrans'] = None if gtr is None else gtr.data_ptr()
        for n, g, gg in zip(GATE_NAMES, gs, ggs):
            weights[n] = g.data_ptr()
            outputs['d' + n] = gg.data_ptr()
        tc.step(weights, x.data_ptr(), lengths.data_ptr(), labels.data_ptr(), B, L, ntok, outputs,
                torch.cuda.current_stream(dev).cuda_stream)
        ctx.has_tr = gtr is not None
        ctx.n_gates = len(gs)
        ctx.save_for_backward(*(grads + ([gtr] if gtr is not None else []) + ggs))
        ctx.mark_non_differentiable(tags)
        return loss.reshape(()), tags

    @staticmethod
    def backward(ctx, gloss, _gtags):
        saved = ctx.saved_tensors
        grads = [g * gloss for g in saved[:7]]
        k = 7
        gtr = None
        if ctx.has_tr:
            gtr = saved[k] * gloss
            k += 1
        ggs = [g * gloss for g in saved[k:k + ctx.n_gates]]
        return (None, None, None, None, None, None) + tuple(grads) + (gtr,) + tuple(ggs)


def decomp_ifst_train_step(tc, Vgen, S1, S2, W, Cmat, h0, hT, P, x, lengths, labels, crf_trans=None, gates=(),
                           valid_tokens=None):
    """Returns (loss scalar tensor with grad, tags int32 [B,L] with -1 at pads).  gates: the tensors Wss1, Wrs1, bs1
    (farnn = 1) followed by Wss2, Wrs2, bs2 (farnn = 2), in that order.  valid_tokens: sum of the clamped lengths if
    the caller already has it (device-resident lengths would otherwise cost a synchronising read per step)."""
    return _DecompIfstTrainStep.apply(tc, valid_tokens, x, lengths, labels, P, Vgen, S1, S2, W, Cmat, h0, hT, crf_trans,
                                      *gates)

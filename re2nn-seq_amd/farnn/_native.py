"""Shared host-side plumbing of the model mirrors: tensors in, C-ABI call, tensors out.

PyTorch-ROCm tensors are used only as device-memory containers; every number is produced by
libfarnn_hip.so.  The method contract mirrors what the reference's callers consume
(SURVEY.md 8b): ``val.val_onehot`` calls ``forward_local(x, label, lengths, train=False)`` and
``RE.get_RE_prediction`` calls ``forward_RE(x, label, lengths, train=False)``.
"""
import os

import numpy as np
import torch

from .. import _lib


def default_device():
    """One process per GPU: LOCAL_RANK picks the device (torch.distributed launch contract)."""
    if 'LOCAL_RANK' in os.environ:
        return int(os.environ['LOCAL_RANK'])
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


class PendingLocal:
    """A forward_local() call whose device work has been enqueued; ``result()`` waits for it and returns the
    reference's triple (None, flat_pred, flat_true)."""

    def __init__(self, owner, ticket, total, true, out_device, dev_result=None):
        self._owner, self._ticket, self._total, self._true = owner, ticket, total, true
        self._out_device, self._dev_result = out_device, dev_result
        self._pred = None
        self._handle = owner._h                           # the device handle this ticket belongs to (never the lazy `handle` property)

    def result(self):
        if self._pred is None:
            if self._dev_result is not None:              # device tensors in: device tensors out
                self._pred = self._dev_result.to(self._out_device)
            else:
                if self._handle is None or self._handle is not self._owner._h:
                    raise _lib.FarnnError('this batch was submitted to a device handle that has since been dropped (invalidate())')
                pred = torch.empty((self._total,), dtype=torch.int64)
                n = self._handle.tag_host_wait(self._ticket, pred.data_ptr())
                assert n == self._total
                self._owner._in_flight -= 1
                self._pred = pred
        return None, self._pred, self._true

    def __del__(self):
        # a ticket dropped without result() (an exception in the caller's loop): wait for its batch so that the library's
        # slot and this model's in-flight count are released.  Only on the handle the batch was submitted to, and only if that
        # handle is still the model's (never build a new one here); a ticket carries its submit's generation, so a finalizer
        # that runs late -- the slot reclaimed and handed to a newer batch meanwhile -- is refused by the library and changes nothing
        if self._pred is None and self._dev_result is None and self._ticket >= 0:
            try:
                h = self._handle
                if h is not None and h is getattr(self._owner, '_h', None):
                    h.tag_host_wait(self._ticket, None)
                    self._owner._in_flight -= 1
            except Exception:
                pass


class NativeTagger:
    """Base of the FARNN_* mirrors.  Subclasses implement ``_build_handle()``."""

    pipeline_depth = 2              # batches in flight on the host-inclusive path (val.val_onehot)

    local_uses_max_len = False      # decomposed models iterate lengths.max() positions

    def __init__(self, args, o_idx, device=None):
        self.args = args
        self.o_idx = int(o_idx)
        self.device_index = default_device() if device is None else int(device)
        self._h = None
        self.training = False

    # ---- nn.Module-like surface the reference drivers touch --------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def cuda(self, device=None):
        if device is not None:
            self.device_index = int(device if not isinstance(device, torch.device) else device.index or 0)
            self.invalidate()
        return self

    def cpu(self):
        raise _lib.FarnnError('the MI355X-native tagger has no CPU path; the CPU oracle lives under '
                              'oracle/ and is test infrastructure only')

    def parameters(self):
        return iter(())

    def invalidate(self):
        """Drop the device handle (call after changing any parameter array)."""
        if self._h is not None:
            self._h.close()
            self._h = None
        self._in_flight = 0

    @property
    def handle(self):
        if self._h is None:
            if not torch.cuda.is_available():
                raise _lib.FarnnError('no MI355X visible (torch.cuda.is_available() is False); '
                                      'the tagging path has no CPU fallback')
            self._h = self._build_handle()
        return self._h

    def _build_handle(self):
        raise NotImplementedError

    # ---- the call --------------------------------------------------------------------------
    def _dev(self):
        return torch.device('cuda', self.device_index)

    def run(self, input, lengths, mode, want_tags=False, want_flat=False, want_scores=False):
        """One farnn_tag() call.  Returns dict(tags[B,L] int32, flat int64[sum len], scores[B,L,K])."""
        dev = self._dev()
        h = self.handle
        x = input.to(device=dev, dtype=torch.int64).contiguous()
        ln = lengths.to(device=dev, dtype=torch.int64).contiguous()
        B, L = x.shape
        out = {}
        tags = flat = scores = None
        if want_tags:
            tags = torch.empty((B, L), dtype=torch.int32, device=dev)
        if want_flat:
            total = int(lengths.detach().cpu().numpy().clip(0, L).sum()) if lengths.device.type == 'cpu' \
                else int(lengths.clamp(0, L).sum().item())
            flat = torch.empty((total,), dtype=torch.int64, device=dev)
        if want_scores:
            scores = torch.empty((B, L, h.num_columns()), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            h.tag(x.data_ptr(), ln.data_ptr(), B, L, mode,
                  None if tags is None else tags.data_ptr(),
                  None if flat is None else flat.data_ptr(),
                  None if scores is None else scores.data_ptr(), stream)
        out['tags'], out['flat'], out['scores'] = tags, flat, scores
        out['_keep'] = (x, ln)
        return out

    @staticmethod
    def _flatten(t, lengths):
        """reference utils.flatten (:153-164) as one masked select.  Host tensors go through numpy: a
        torch CPU masked select spins up the intra-op thread pool (128 threads on the GPU box), whose
        spinning workers then slow the HIP runtime's own threads -- 2 ms per call inside the tagging loop
        against 14 us for numpy."""
        L = t.shape[1]
        if t.device.type == 'cpu':
            ln = lengths.detach().cpu().numpy()
            mask = np.arange(L)[None, :] < ln[:, None]
            return torch.from_numpy(t.detach().numpy()[mask])
        mask = torch.arange(L, device=t.device)[None, :] < lengths.to(t.device)[:, None]
        return t[mask]

    def _clip_len(self, input, lengths):
        if self.local_uses_max_len:
            return input[:, :int(lengths.max().item())]
        return input

    # ---- host-inclusive path: CPU tensors in, CPU tensors out, batches in flight ------------------
    def submit_local(self, input, label, lengths):
        """Enqueue one forward_local(train=False) call and return a PendingLocal.  CPU inputs go through the library's
        host-buffer entry (farnn_tag_host_submit: pinned staging, the H2D copies, the tagging launch and the D2H copy of
        the flat predictions on three event-chained streams), so the caller can prepare and submit the next batch
        while this one runs -- the reference's eval loop (val.py:17-31) with batches in flight instead of a synchronous
        H2D / step / D2H round trip per batch (r01: 143 us per batch against 55 us of device time).  Results must be
        collected in submission order."""
        input = self._clip_len(input, lengths)
        B, L = input.shape
        if input.device.type == 'cpu':
            if getattr(self, '_in_flight', 0) >= _lib.HOST_SLOTS:
                raise _lib.FarnnError('more than {} batches in flight: collect the oldest result first'.format(_lib.HOST_SLOTS))
            x = input if (input.dtype == torch.int64 and input.is_contiguous()) else input.to(torch.int64).contiguous()
            ln = lengths if (lengths.dtype == torch.int64 and lengths.is_contiguous()) else lengths.to(torch.int64).contiguous()
            ticket, total = self.handle.tag_host_submit(x.data_ptr(), ln.data_ptr(), B, L)
            self._in_flight = getattr(self, '_in_flight', 0) + 1
            # the flat gold labels (utils.flatten): host work that overlaps the device's
            if label.device.type == 'cpu' and label.dtype == torch.int64 and label.is_contiguous() and label.shape[1] == L:
                true = torch.empty((total,), dtype=torch.int64)
                _lib.flatten_host(label.data_ptr(), ln.data_ptr(), B, L, true.data_ptr())
            else:
                true = self._flatten(label, lengths)
            return PendingLocal(self, ticket, total, true, input.device)
        r = self.run(input, lengths, _lib.MODE_LOCAL, want_flat=True)
        return PendingLocal(self, -1, r['flat'].shape[0], self._flatten(label, lengths).to(input.device), input.device,
                            dev_result=r['flat'])

    # ---- reference method contract ---------------------------------------------------------
    def forward_local(self, input, label, lengths, train=True, re_tags=None):
        """(loss, flat_pred int64[sum len], flat_true int64[sum len]); loss is None.
        Reference: model_onehot.py:131-146 / model_decompose_single.py:207-304."""
        if train:
            raise NotImplementedError('training (loss/backward) is outside the forward tagging path; '
                                      'call forward_local(..., train=False)')
        return self.submit_local(input, label, lengths).result()

    def forward_score(self, input, label, lengths, train=True):
        """Unclamped scores [B,L,K] for all L positions (model_onehot.py:351-428)."""
        r = self.run(input, lengths, _lib.MODE_FULL, want_scores=True)
        return r['scores'].to(input.device)

    def forward_RE(self, input, label, lengths, train=False):
        """(pred[B,L] int64, scores[B,L,K] with the `oo` column clamped); pads included
        (model_onehot.py:148-160)."""
        r = self.run(input, lengths, _lib.MODE_RE, want_tags=True, want_scores=True)
        return r['tags'].to(torch.int64).to(input.device), r['scores'].to(input.device)

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


class NativeTagger:
    """Base of the FARNN_* mirrors.  Subclasses implement ``_build_handle()``."""

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

    # ---- reference method contract ---------------------------------------------------------
    def forward_local(self, input, label, lengths, train=True, re_tags=None):
        """(loss, flat_pred int64[sum len], flat_true int64[sum len]); loss is None.
        Reference: model_onehot.py:131-146 / model_decompose_single.py:207-304."""
        if train:
            raise NotImplementedError('training (loss/backward) is outside the forward tagging path; '
                                      'call forward_local(..., train=False)')
        r = self.run(self._clip_len(input, lengths), lengths, _lib.MODE_LOCAL, want_flat=True)
        pred = r['flat'].to(input.device)
        true = self._flatten(label, lengths).to(input.device)
        return None, pred, true

    def forward_score(self, input, label, lengths, train=True):
        """Unclamped scores [B,L,K] for all L positions (model_onehot.py:351-428)."""
        r = self.run(input, lengths, _lib.MODE_FULL, want_scores=True)
        return r['scores'].to(input.device)

    def forward_RE(self, input, label, lengths, train=False):
        """(pred[B,L] int64, scores[B,L,K] with the `oo` column clamped); pads included
        (model_onehot.py:148-160)."""
        r = self.run(input, lengths, _lib.MODE_FULL, want_tags=True, want_scores=True)
        scores = r['scores']
        K = scores.shape[2]
        scores[:, :, K - 1].clamp_(max=float(self.args.threshold))
        return r['tags'].to(torch.int64).to(input.device), scores.to(input.device)

"""Soak of the one-launch tagging step's inter-workgroup hand-off (csrc/chain_regs.hip.h), collected by `pytest -m gpu`.

>= 1e5 launches of the fused kernel (the recurrence with the scores + decode beside it: the two workgroups of a sequence
exchange stash rows through the progress / arrival words) while a second stream saturates HBM and LDS with copies and
matrix products, so that workgroups of different kernels share compute units, the fused workgroups are pre-empted for
issue slots unevenly and L2 lines get evicted (MI355X_MICROARCH.md: "test every hand-off under UNEVEN load, consumer
L1-warm, checking every word").  B in {8, 256, 1024}; LOCAL and FULL mode; consecutive launches alternate between
different batches on the same handle, so a stale stash line of the previous launch -- same address -- would decode into
the previous batch's tags.  EVERY launch's tags are compared on the device with the two-kernel form of the same step
(FARNN_NOFUSE=1: recurrence kernel + score kernel, no hand-off), which the parity suite holds to the oracle.

FARNN_SOAK_SCALE multiplies the launch counts (default 1 -> 1e5 launches, ~15 s).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


class _Noise:
    """keeps a side stream busy: 64 MiB copies (HBM) and 1024^3 fp32 matrix products (LDS + matrix cores)"""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        with torch.cuda.stream(self.stream):
            self.a = torch.randn(16 << 20, device='cuda')
            self.b = torch.empty_like(self.a)
            self.m = torch.randn(1024, 1024, device='cuda')
            self.n = torch.empty_like(self.m)
        self.i = 0

    def kick(self):
        with torch.cuda.stream(self.stream):
            if self.i % 3 == 2:
                torch.mm(self.m, self.m, out=self.n)
            else:
                self.b.copy_(self.a)
        self.i += 1


def _soak(B, L, launches, seed, full, S=71):
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(seed)
    V, C = 950, 128
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
    os.environ['FARNN_FUSE'] = '1'                         # the one-launch form at every batch size (round 5: the default takes two
    try:                                                   # launches once 2 B exceeds the compute units -- the hand-off is what is soaked here)
        h = _lib.create_onehot_ifst(T, W, O, h0, hT, o_idx=0)
    finally:
        del os.environ['FARNN_FUSE']
    mode = _lib.MODE_FULL if full else _lib.MODE_LOCAL
    nb = 6
    batches = []
    os.environ['FARNN_NOFUSE'] = '1'                       # the reference form: two kernels, no hand-off (the switches are read
    try:                                                   # when a handle is created: a second handle of the same model)
        href = _lib.create_onehot_ifst(T, W, O, h0, hT, o_idx=0)
        for k in range(nb):
            x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
            if k == 1:
                lengths[:] = L                             # every chain full length
            if k == 2:
                lengths[:] = np.maximum(1, lengths // 4)   # short chains: everything is "the last tile"
            xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
            ref = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
            href.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, ref.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert 'fused' not in href.kernel_name(_lib.KERN_CHAIN)
            batches.append((xd, ld, ref))
        href.close()
    finally:
        del os.environ['FARNN_NOFUSE']
    noise = _Noise()
    outs = [torch.full((B, L), -7, dtype=torch.int32, device='cuda') for _ in range(4)]
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    for i in range(launches):
        xd, ld, ref = batches[(i * 5 + i // 7) % nb]
        out = outs[i % 4]
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, out.data_ptr(), None, None, s)
        bad += (out != ref).sum()                          # on the device, behind the launch on the same stream
        if i % 4 == 0:
            noise.kick()
        if i % 2048 == 2047:
            torch.cuda.synchronize()                       # bounds the queued work; the bursts in between are unsynchronised
    torch.cuda.synchronize()
    assert 'fused' in h.kernel_name(_lib.KERN_CHAIN)       # the launches above really were the one-launch form
    h.close()
    return int(bad.item())


@pytest.mark.parametrize('B,L,launches,full', [(8, 64, 40000, False), (256, 64, 30000, False), (256, 64, 10000, True),
                                              (1024, 64, 6000, False), (300, 100, 8000, False), (64, 33, 6000, True)])
def test_fused_handoff_soak(B, L, launches, full):
    scale = float(os.environ.get('FARNN_SOAK_SCALE', '1'))
    n = max(64, int(launches * scale))
    bad = _soak(B, L, n, seed=B * 131 + L + int(full), full=full)
    assert bad == 0, '{} wrong tags over {} fused launches (B={}, L={}, full={})'.format(bad, n, B, L, full)


@pytest.mark.parametrize('S,B,L,launches,full', [(104, 256, 64, 12000, False), (104, 8, 64, 12000, False), (104, 301, 50, 5000, True),
                                                (128, 256, 64, 5000, False), (97, 64, 33, 5000, False)])
def test_fused_handoff_soak_wide_form(S, B, L, launches, full):
    """The same soak for round 4's wide form of the recurrence (chain_wide.hip.h: 72 < S <= 128; S = 104: two workgroups per
    compute unit, S = 128: one) -- the same progress / arrival words, the device-side epoch, other code around them."""
    scale = float(os.environ.get('FARNN_SOAK_SCALE', '1'))
    n = max(64, int(launches * scale))
    bad = _soak(B, L, n, seed=S * 977 + B * 131 + L + int(full), full=full, S=S)
    assert bad == 0, '{} wrong tags over {} fused launches (S={}, B={}, L={}, full={})'.format(bad, n, S, B, L, full)

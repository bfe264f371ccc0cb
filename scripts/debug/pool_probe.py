#!/usr/bin/env python3
"""What bounds the dense recurrence kernel's step: the same 256 x 64 batch shape with the word ids drawn from pools of different
sizes (1 word: its two 20 KB blocks sit in every compute unit's L1; 16 words: 0.65 MB, L2-resident in every XCD; 950 words uniform:
38.9 MB, mostly Infinity-Cache; the bench's own Zipf draw), ragged and full-length.  Per pool: kernel time by events over 200
launches, bytes the kernel's lanes ask for per compute unit and clock.

    python scripts/debug/pool_probe.py            (FARNN_NOFUSE etc. apply as usual)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from re2nn_seq_amd import _lib  # noqa: E402

B, L = 256, 64
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
h, x0, len0, _ = bench.build_workload('ifst', B, L, 0, 50, False)
h.reserve(B, L)
V, S = 950, 71
rng = np.random.RandomState(7)
stream = torch.cuda.current_stream(dev).cuda_stream
for full in (False, True):
    lengths = np.full(B, L, np.int64) if full else len0
    for name, pool in (('bench draw (Zipf over 949 words)', None), ('1 word', 1), ('4 words', 4), ('16 words', 16), ('64 words', 64), ('256 words', 256), ('949 words, uniform', 949)):
        x = x0.copy() if pool is None else rng.randint(0, pool, size=(B, L)).astype(np.int64)
        if pool is not None or full:
            for b in range(B):
                x[b, lengths[b]:] = V - 1
        if pool is None and full:
            x = rng.zipf(1.1, size=(B, L)) % (V - 1)
        xd, ld = torch.from_numpy(np.ascontiguousarray(x)).to(dev), torch.from_numpy(lengths).to(dev)
        tags = torch.empty((B, L), dtype=torch.int32, device=dev)
        for _ in range(20):
            h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n):
            h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None, stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        toks = int(lengths.sum())
        print('%-8s %-34s %7.2f us per step (both launches), %5.2f TB/s of block bytes' % (
            'full' if full else 'ragged', name, us, toks * 2 * S * S * 4 / (us * 1e-6) / 1e12), flush=True)

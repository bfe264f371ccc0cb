#!/usr/bin/env python3
"""Host time of one farnn_tag call (the enqueue, no synchronisation) against the GPU time of the step: how far the host is from being
the bound.  python scripts/debug/host_enqueue.py [workload] [--states N]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from re2nn_seq_amd import _lib  # noqa: E402

B, L = 256, 64
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
for name, crf in (('ifst', False), ('ifst_crf', True)):
    h, x, lengths, _ = bench.build_workload(name, B, L, 0, 50, False)
    h.reserve(B, L)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev)
    tags = torch.empty((B, L), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    mode = _lib.MODE_CRF if crf and hasattr(_lib, 'MODE_CRF') else _lib.MODE_LOCAL
    for _ in range(50):
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, tags.data_ptr(), None, None, s)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, tags.data_ptr(), None, None, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-10s host enqueue %.1f us per call (python + ctypes + the library), all %d steps done after %.1f us per step' % (
        name, (t1 - t0) / n * 1e6, n, (t2 - t0) / n * 1e6))

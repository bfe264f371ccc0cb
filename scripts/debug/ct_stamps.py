#!/usr/bin/env python3
"""Per-workgroup life of ONE compact_tag_kernel launch (profiling build, FARNN_DBG=2048):

    FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=2048 python scripts/debug/ct_stamps.py [--full-length]

The kernel leaves {seq, len, start, set-up done, xcc, se, cu, -, end of wavefront 0..7} (100 MHz wall clock) per workgroup in a
device buffer; this prints how far the starts are apart, which wavefront holds a workgroup and how the workgroups share units."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from re2nn_seq_amd import _lib  # noqa: E402

B, L = 256, 64
full = '--full-length' in sys.argv
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
h, x, lengths, _ = bench.build_workload('ifst', B, L, 0, 50, full)
h.reserve(B, L)
h.set_compact(True)
xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev)
tags = torch.empty((B, L), dtype=torch.int32, device=dev)
for _ in range(20):
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((B, 16), dtype=np.int64)
rc = lib.farnn_debug_ct_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(B))
assert rc == 0, rc
seq, ln, st, su, xcc, se, cu = buf.T[:7]
ends = buf[:, 8:16]
t0 = st.min()
us = lambda t: (t - t0) * 1e-2
en = ends.max(axis=1)
print('kernel: %s; %d workgroups, starts within %.2f us, last end +%.2f us' % (h.kernel_name(_lib.KERN_CHAIN), B, us(st).max(), us(en).max()))
print('  set-up: mean %.2f max %.2f us' % (((su - st) * 1e-2).mean(), ((su - st) * 1e-2).max()))
for lo, hi in ((1, 15), (15, 25), (25, 35), (35, 45), (45, 55), (55, 64), (64, 65)):
    m = (ln >= lo) & (ln < hi)
    if m.sum():
        e = (ends[m] - st[m, None]) * 1e-2
        print('  len [%2d,%2d): n=%3d start mean +%.2f max +%.2f; life mean %.2f max %.2f us; end of wavefronts 0..7 after the start (mean): %s' % (
            lo, hi, m.sum(), us(st[m]).mean(), us(st[m]).max(), (en[m] - st[m]).mean() * 1e-2, (en[m] - st[m]).max() * 1e-2,
            ' '.join('%.1f' % v for v in e.mean(axis=0))))
key = xcc * 10000 + se * 100 + cu
per = np.bincount(np.unique(key, return_inverse=True)[1])
print('  compute units used: %d, workgroups per unit min %d max %d, histogram %s' % (len(per), per.min(), per.max(), np.bincount(per).tolist()))
for i in np.argsort(en)[-6:][::-1]:
    mates = [j for j in np.where(key == key[i])[0] if j != i]
    print('  last: wg %3d seq %3d len %2d start +%.2f end +%.2f (life %.2f us); wavefront ends %s; shares its unit with %s' % (
        i, seq[i], ln[i], us(st[i]), us(en[i]), (en[i] - st[i]) * 1e-2, ' '.join('%.1f' % v for v in (ends[i] - st[i]) * 1e-2),
        ', '.join('wg %d (len %d, +%.2f..+%.2f)' % (j, ln[j], us(st[j]), us(en[j])) for j in mates) or 'nobody'))

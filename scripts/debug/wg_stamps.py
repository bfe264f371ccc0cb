#!/usr/bin/env python3
"""Per-workgroup life of ONE headline launch (profiling build, FARNN_DBG=2048):

    FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=2048 [FARNN_NODEST=1] [FARNN_NOFUSE=1] python scripts/debug/wg_stamps.py [--full-length]

chain_regs_kernel's wrapper leaves {seq, len, start, end (100 MHz wall clock), shader cycles, xcc, se, cu} per workgroup in a device
buffer; this script runs the bench's batch a few times, reads the buffer of the last launch back and prints who ends last, the
life of the workgroups by sequence length and what shares a compute unit with the stragglers."""
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
xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev)
tags = torch.empty((B, L), dtype=torch.int32, device=dev)
for _ in range(20):
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
lib = _lib.load()
n = 2 * B
buf = np.zeros((n, 16), dtype=np.int64)
rc = lib.farnn_debug_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n))
assert rc == 0, rc
seq, ln, st, en, cyc, xcc, se, cu = buf.T[:8]
c_setup, c_chain, c_scorer, c_meet, c_tiles = buf.T[8:13]
t0 = st.min()
start, end, life = (st - t0) * 1e-2, (en - t0) * 1e-2, (en - st) * 1e-2          # us
clk = cyc / np.maximum((en - st) * 10.0, 1)
print('kernel: %s; %d workgroups, starts within %.2f us, last end +%.2f us, clock median %.3f GHz' % (
    h.kernel_name(_lib.KERN_CHAIN), n, start.max(), end.max(), np.median(clk)))
key = xcc * 10000 + se * 100 + cu
for i in np.argsort(end)[-6:][::-1]:
    mates = [j for j in np.where(key == key[i])[0] if j != i]
    print('  last: wg %3d seq %3d dir %d len %2d  start +%.2f end +%.2f (life %.2f us, %d cycles: setup %d, chain %d = %d per step, scorer done %+d, tiles + arrival %d)  shares its compute unit with %s' % (
        i, seq[i], i & 1, ln[i], start[i], end[i], life[i], cyc[i], c_setup[i], c_chain[i], c_chain[i] // max(ln[i], 1), c_scorer[i], c_tiles[i],
        ', '.join('wg %d (len %d, end +%.2f)' % (j, ln[j], end[j]) for j in mates) or 'nobody'))
for lo, hi in ((1, 15), (15, 25), (25, 35), (35, 45), (45, 55), (55, 64), (64, 65)):
    m = (ln >= lo) & (ln < hi)
    if m.sum():
        print('  len [%2d,%2d): n=%3d  life mean %.2f max %.2f us; end mean +%.2f max +%.2f; cycles: setup %.0f, chain %.0f (%.0f per step), tiles + arrival %.0f (max %d)' % (
            lo, hi, m.sum(), life[m].mean(), life[m].max(), end[m].mean(), end[m].max(), c_setup[m].mean(), c_chain[m].mean(),
            (c_chain[m] / np.maximum(ln[m], 1)).mean(), c_tiles[m].mean(), c_tiles[m].max()))
per = np.bincount(np.unique(key, return_inverse=True)[1])
print('  compute units used: %d, workgroups per unit min %d max %d' % (len(per), per.min(), per.max()))
# pairs on a unit: how the lengths are paired
pairs = {}
for i in range(n):
    pairs.setdefault(key[i], []).append(i)
tot = sorted(((max(end[v]) , sorted(ln[v].tolist())) for v in pairs.values()), reverse=True)[:8]
print('  units that finish last (end, lengths of their workgroups):', ['+%.2f %s' % t for t in tot])

#!/usr/bin/env python3
"""Summarise FARNN_DBG=2048 output of the profiling build (chain_regs_kernel's wrapper: one line per workgroup with its start / end
on the 100 MHz wall clock, its shader cycles and the compute unit it ran on):  python scripts/wg_lifetimes.py file..."""
import re
import sys
import numpy as np

PAT = re.compile(r'wg (\d+) seq (-?\d+) dir (\d) len (-?\d+): start (\d+) end (\d+) \(10 ns ticks\), (\d+) cycles, xcc (\d+) se (\d+) cu (\d+)')
for f in sys.argv[1:]:
    a = np.array([[int(v) for v in m.groups()] for m in map(PAT.match, open(f)) if m])
    print(f, len(a), 'workgroup lines')
    a = a[np.argsort(a[:, 4])]
    groups = np.split(a, np.where(np.diff(a[:, 4]) > 2000)[0] + 1)
    for g in groups[-2:]:
        t0 = g[:, 4].min()
        dur = (g[:, 5] - g[:, 4]) * 10e-3
        end = (g[:, 5] - t0) * 10e-3
        st = (g[:, 4] - t0) * 10e-3
        clk = g[:, 6] / ((g[:, 5] - g[:, 4]) * 10.0)
        print('  launch of %d workgroups: starts within %.2f us, last end +%.2f us, clock median %.3f GHz' % (len(g), st.max(), end.max(), np.median(clk)))
        for i in np.argsort(end)[-4:][::-1]:
            print('     last: wg %3d seq %3d dir %d len %2d start +%.2f end +%.2f (%.2f us, %d cycles) xcc %d se %d cu %d' % (
                g[i, 0], g[i, 1], g[i, 2], g[i, 3], st[i], end[i], dur[i], g[i, 6], g[i, 7], g[i, 8], g[i, 9]))
        for lo, hi in ((5, 15), (15, 25), (25, 35), (35, 45), (45, 55), (55, 64), (64, 65)):
            m = (g[:, 3] >= lo) & (g[:, 3] < hi)
            if m.sum():
                print('     len [%2d,%2d): n=%3d  mean life %.2f us, mean end +%.2f, max end +%.2f' % (lo, hi, m.sum(), dur[m].mean(), end[m].mean(), end[m].max()))
        # who shares a compute unit: (xcc, se, cu) -> the workgroups
        key = g[:, 7] * 10000 + g[:, 8] * 100 + g[:, 9]
        n_per = np.bincount(np.unique(key, return_inverse=True)[1])
        print('     compute units used: %d; workgroups per unit: min %d max %d' % (len(n_per), n_per.min(), n_per.max()))

"""Host-to-host rate of the boundary (reference calling convention: CPU tensors in, CPU tensors out through
FARNN_S_O_I_S.forward_local): includes the PCIe copies of x / lengths / tags and the Python glue."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from util import ns                                   # noqa: E402
from re2nn_seq_amd import synth                       # noqa: E402
from re2nn_seq_amd.farnn.model_onehot import FARNN_S_O_I_S   # noqa: E402

rng = np.random.RandomState(1234)
V, S, C, B, L = 950, 71, 129, 256, 64
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))
m = FARNN_S_O_I_S(T, O, W, np.zeros(S), hT, h0, None, ns(), o_idx=0)
xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
lab = torch.zeros_like(xt)
for _ in range(20):
    m.forward_local(xt, lab, lt, train=False)
n = 300
t0 = time.perf_counter()
for _ in range(n):
    _, pred, _ = m.forward_local(xt, lab, lt, train=False)
el = time.perf_counter() - t0
print('host-inclusive, one synchronous call per batch: {:.3e} valid tokens/s, {:.1f} us per batch ({} valid tokens)'.format(
    int(lengths.sum()) * n / el, el / n * 1e6, int(lengths.sum())))
import collections
for depth in (1, 2, 3):
    m.pipeline_depth = depth
    q = collections.deque()
    t0 = time.perf_counter()
    for _ in range(n):
        q.append(m.submit_local(xt, lab, lt))
        if len(q) > depth:
            q.popleft().result()
    while q:
        q.popleft().result()
    el = time.perf_counter() - t0
    print('host-inclusive, {} batches in flight: {:.3e} valid tokens/s, {:.1f} us per batch'.format(
        depth, int(lengths.sum()) * n / el, el / n * 1e6))

if os.environ.get('FARNN_HOST_PROFILE'):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    q = collections.deque()
    for _ in range(200):
        q.append(m.submit_local(xt, lab, lt))
        if len(q) > 2:
            q.popleft().result()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)

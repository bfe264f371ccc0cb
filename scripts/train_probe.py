"""Times the HIP training step at the SNIPS-sized decomposed configuration (probe; bench.py --workload train is the
reported form)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from re2nn_seq_amd import _lib, synth  # noqa: E402
from re2nn_seq_amd.farnn.train_step import decomp_ifst_train_step  # noqa: E402

V, S, R, K, B, L, D = 11000, 104, 50, 73, 256, 64, 100
rng = np.random.RandomState(0)
dev = torch.device('cuda')
f = lambda *shape, sc=0.3: torch.from_numpy((rng.randn(*shape) * sc).astype(np.float32)).to(dev).requires_grad_(True)  # noqa: E731
Cm = np.zeros((K, S), np.float32)
Cm[rng.randint(0, K, size=S), np.arange(S)] = 1
p = dict(S1=f(S, R, sc=0.1), S2=f(S, R, sc=0.1), V_embed=f(V, R, sc=0.8), G=f(D, R), E=f(V, D),
         C=torch.from_numpy(Cm).to(dev).requires_grad_(True),
         W=torch.from_numpy(((rng.rand(S, S) < 1.0 / S) * 0.5).astype(np.float32)).to(dev).requires_grad_(True),
         h0=f(S, sc=0.5), hT=f(S, sc=0.5))
beta = torch.full((R,), 0.7, device=dev)
x, lengths = synth.random_batch(V, B, L, rng)
labels = rng.randint(0, K, size=(B, L)).astype(np.int64)
xd, ld, lab = torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev), torch.from_numpy(labels).to(dev)
tc = _lib.TrainContext(V, S, R, K, nl='tanh', threshold=0.5, o_idx=0)
opt = torch.optim.Adam(list(p.values()), lr=1e-3)
tc.set_profiling(1)


def step():
    opt.zero_grad(set_to_none=True)
    Vgen = p['V_embed'] * beta + torch.tanh(p['E'] @ p['G']) * (1 - beta)
    loss, _ = decomp_ifst_train_step(tc, Vgen, p['S1'], p['S2'], p['W'], p['C'], p['h0'], p['hT'], None, xd, ld, lab)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
tc.time()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / n
ms, k = tc.time()
print('tokens', int(lengths.sum()), 'step %.3f ms (library part %.3f ms)' % (el * 1e3, ms / max(k, 1)),
      'tokens/s %.3e' % (lengths.sum() / el), 'loss', float(loss.detach()))

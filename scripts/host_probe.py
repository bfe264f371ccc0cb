import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from re2nn_seq_amd import synth, _lib
rng = np.random.RandomState(1234)
V, S, C, B, L = 950, 71, 129, 256, 64
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))
h = _lib.create_onehot_ifst(T, W, O, h0, hT)
xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def T_(f, n=50):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
print('tag(tags) host us', T_(lambda: h.tag(xd.data_ptr(), ld.data_ptr(), B, L, 0, tags.data_ptr(), None, None, st)))
print('tag(flat) host us', T_(lambda: h.tag(xd.data_ptr(), ld.data_ptr(), B, L, 0, None, flat.data_ptr(), None, st)))
print('tag+sync us', T_(lambda: (h.tag(xd.data_ptr(), ld.data_ptr(), B, L, 0, None, flat.data_ptr(), None, st), torch.cuda.synchronize())))
xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
print('x.to(cuda) us', T_(lambda: xt.to('cuda')))
print('flat.cpu() us', T_(lambda: flat.cpu()))
lab = torch.zeros_like(xt)
def fl():
    mask = torch.arange(L)[None, :] < lt[:, None]
    return lab[mask]
print('torch flatten us', T_(fl))
def fl2():
    mask = np.arange(L)[None, :] < lengths[:, None]
    return torch.from_numpy(lab.numpy()[mask])
print('numpy flatten us', T_(fl2))
print('threads', torch.get_num_threads())

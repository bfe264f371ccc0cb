"""one launch (FARNN_FUSE=1) against two (FARNN_NOFUSE=1) for the onehot i-FST at ATIS size over a grid of batch sizes and sequence
lengths: the data the dispatch rule of launch_chain (csrc/farnn_hip.hip) is calibrated on (profiles/r06_dispatch_grid.txt)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from re2nn_seq_amd import _lib, synth
V, S, C = 950, 71, 128
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, np.random.RandomState(1234))
hs = {}
for name, env in (('one', 'FARNN_FUSE'), ('two', 'FARNN_NOFUSE')):
    os.environ[env] = '1'
    hs[name] = _lib.create_onehot_ifst(T, W, O, h0, hT)
    del os.environ[env]
def t(h, xd, ld, B, L, tags, n=300):
    for _ in range(30): h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('B    L    one launch   two launches   (us per step, best of 3 interleaved rounds; lengths U[5, L] with one full-length row)')
for B in (16, 32, 64, 96, 128):
    for L in (16, 30, 40, 48, 56, 64, 100):
        x, lengths = synth.random_batch(V, B, L, np.random.RandomState(99))
        xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
        tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
        r = {k: min(t(h, xd, ld, B, L, tags) for _ in range(3)) for k, h in hs.items()}
        print('{:<4d} {:<4d} {:9.1f}   {:9.1f}      {}'.format(B, L, r['one'], r['two'], 'one' if r['one'] < r['two'] else 'two'))

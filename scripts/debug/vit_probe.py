"""Phase cycle counts (FARNN_DBG=8192) and kernel time of the fused Viterbi kernel on the config-3 batch (diagnostic).
    python scripts/debug/vit_probe.py [C] [lib-suffix]      C = label columns (K = C + 2), lib-suffix e.g. _b"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from re2nn_seq_amd import _lib, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
if len(sys.argv) > 2:
    _lib.LIB_PATH = _lib.LIB_PATH.replace('.so', sys.argv[2] + '.so')
rng = np.random.RandomState(1234)
V, S, B, L = 950, 71, 256, 64
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))
K = C + 2
tr = (np.random.RandomState(3).randn(K, K) * 0.1).astype(np.float32)
h = _lib.create_onehot_ifst(T, W, O, h0, hT, o_idx=0, use_crf=True, crf_trans=tr)
xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
def run(n):
    for _ in range(n):
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
    torch.cuda.synchronize()
run(20)
t0 = time.perf_counter(); run(300); el = (time.perf_counter() - t0) / 300
print('K = {} lib {}: {:.1f} us per step (chain + viterbi)'.format(K, os.path.basename(_lib.LIB_PATH), el * 1e6))
if os.environ.get('VIT_PROBE', '1') != '0':
    os.environ['FARNN_DBG'] = '8192'
    run(1)

"""Timing of the decomposed recurrence at shapes bench.py has no flag for (diagnostic).  python scripts/debug/rows_shapes.py S R farnn"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from re2nn_seq_amd import _lib, synth
import test_gpu_parity_bench_size as tb
S, R, farnn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
V, q, gates, tr = tb._snips_model(R, farnn, False, S=S)
B, L = 256, 64
x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))
h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn, gates=gates,
                            sigmoid_exponent=5, nl='tanh', threshold=0.5, o_idx=0)
xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
def run(n):
    for _ in range(n):
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
    torch.cuda.synchronize()
run(5)
t0 = time.perf_counter(); run(50); el = (time.perf_counter() - t0) / 50
print('S={} R={} farnn={} NOREGS={}: {:.1f} us per step, {:.3e} tokens/s'.format(S, R, farnn, os.environ.get('FARNN_ROWS_NOREGS', ''), el * 1e6, lengths.sum() / el))

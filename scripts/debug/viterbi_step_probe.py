"""Forward-pass / back-trace cycles per step of the CRF decode for a tag count (profiling build: FARNN_LIB=...probes.so FARNN_DBG=8192).
    python scripts/debug/viterbi_step_probe.py C [S]          (K = C + 2 tags)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from re2nn_seq_amd import _lib, synth                                             # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle'))

C = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else 71
rng = np.random.RandomState(5)
V, B, L = 300, 256, 64
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=8.0)
tr = np.zeros((C + 2, C + 2), np.float32); tr[:, C] = -1e4; tr[C + 1, :] = -1e4
tr += rng.randn(C + 2, C + 2).astype(np.float32) * 0.1
h = _lib.create_onehot_ifst(T, W, O, h0, hT, o_idx=0, use_crf=True, crf_trans=tr)
x, lengths = synth.random_batch(V, B, L, rng, min_len=L)
xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
for _ in range(2):
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
ev[1].record(); torch.cuda.synchronize()
print('C = %d (K = %d), S = %d: %.1f us per step, kernel %s' % (C, C + 2, S, ev[0].elapsed_time(ev[1]) * 50, h.kernel_name(_lib.KERN_CHAIN)))

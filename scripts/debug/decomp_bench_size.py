"""debug: decomposed i-FST at bench size vs the oracle, per-sequence error pattern"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import farnn_oracle as fo
from re2nn_seq_amd import _lib, synth
import test_gpu_parity_bench_size as t

R, farnn, crf = int(sys.argv[1]), int(sys.argv[2]), False
B, L = int(sys.argv[3]), 64
V, q, gates, tr = t._snips_model(R, farnn, crf)
x, lengths = synth.random_batch(V, 256, L, np.random.RandomState(4321))
x, lengths = x[:B], lengths[:B]
h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn, gates=gates,
                            sigmoid_exponent=5, nl='tanh', threshold=0.5, o_idx=0)
print(h.kernel_name(0))
xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
K = q['Cout'].shape[0]
sc = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, None, sc.data_ptr())
torch.cuda.synchronize()
ref = fo.decomp_ifst_scores(q, x, lengths)
Lm = ref.shape[1]
got = sc.cpu().numpy()[:, :Lm]
m = np.arange(Lm)[None, :] < lengths[:, None]
err = np.abs(got - ref).max(axis=2) * m
print('max err', err.max(), 'bad seqs', (err.max(1) > 1e-4).sum(), 'of', B)
bad = np.where(err.max(1) > 1e-4)[0]
for b in bad[:12]:
    pos = np.where(err[b] > 1e-4)[0]
    print(' seq', b, 'len', lengths[b], 'bad positions', pos[:10], '... n=', len(pos), 'maxerr %.4f' % err[b].max())

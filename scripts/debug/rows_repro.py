import os, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import farnn_oracle as fo
from re2nn_seq_amd import _lib, synth
S, R, farnn, C, B, L = [int(v) for v in sys.argv[1:7]]
rng = np.random.RandomState(5)
f = lambda a: np.asarray(a, np.float32)
V = 300
p = synth.random_decomposed_params(V, S, C, R, 20, rng, contractive=True)
q = {'Vgen': f(p['V_embed']), 'S1': f(p['S1']), 'S2': f(p['S2']), 'W': f(p['wildcard_mat']), 'Cout': f(p['C_output_mat']),
     'h0': f(p['start_vector']), 'hT': f(p['final_vector']), 'farnn': farnn, 'nl': fo.NL_TANH, 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
gates = {'Wss1': f(rng.randn(S, S) * 0.03), 'Wrs1': f(rng.randn(R, S) * 0.03), 'bs1': f(np.full(S, 1.0))}
if farnn == 2:
    gates.update(Wss2=f(rng.randn(S, S) * 0.03), Wrs2=f(rng.randn(R, S) * 0.03), bs2=f(np.full(S, 1.0)))
q.update(gates)
x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
K = q['Cout'].shape[0]
h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn, gates=gates,
                            sigmoid_exponent=5, nl='tanh', threshold=0.5, o_idx=0)
scores = torch.full((B, L, K), np.nan, dtype=torch.float32, device='cuda')
tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, scores.data_ptr())
torch.cuda.synchronize()
print('ok', h.kernel_name(_lib.KERN_CHAIN), os.environ.get('FARNN_ROWS_NOROUNDS'), float(np.nanmax(np.abs(scores.cpu().numpy()))))

"""Ad-hoc soak (run on the GPU box; test infrastructure -- it uses the oracle): random ragged batches through
the fused-Viterbi and the decomposed rows paths, each compared with the numpy oracle.
Usage: python tests/soak_crf_decomp.py [iters]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import farnn_oracle as fo          # noqa: E402
from re2nn_seq_amd import _lib, synth         # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(77)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731

# ---- CRF (fused scores + Viterbi), K = 66 tags
V, S, C = 300, 41, 64
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=6.0)
tr = fo.crf_default_transitions(C) + rng.randn(C + 2, C + 2).astype(np.float32)
h = _lib.create_onehot_ifst(T, W, O, h0, hT, use_crf=True, crf_trans=tr, o_idx=3)
bad = 0
for it in range(iters):
    B = int(rng.choice([1, 3, 17, 64])); L = int(rng.choice([1, 7, 30, 64]))
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    flat = torch.full((int(lengths.sum()),), -7, dtype=torch.int64, device='cuda')
    h.tag(t(x).data_ptr(), t(lengths).data_ptr(), B, L, _lib.MODE_LOCAL, None, flat.data_ptr(), None)
    xd, ld = t(x), t(lengths)
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, flat.data_ptr(), None)
    torch.cuda.synchronize()
    sc = fo.onehot_crf_extension_scores(fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths))
    ref = fo.forward_local_tags(sc, lengths, 0.5, 3, crf_tr=tr)
    if not np.array_equal(flat.cpu().numpy(), ref):
        bad += 1
        print('CRF mismatch at', it, B, L)
print('crf: {} iterations, {} mismatches'.format(iters, bad))

# ---- decomposed rows kernel, gated
V, S, K, R = 200, 60, 9, 80
p = synth.random_decomposed_params(V, S, K, R, 16, rng)
q = {'Vgen': p['V_embed'].astype(np.float32), 'S1': p['S1'].astype(np.float32), 'S2': p['S2'].astype(np.float32),
     'W': p['wildcard_mat'].astype(np.float32), 'Cout': p['C_output_mat'].astype(np.float32),
     'h0': p['start_vector'].astype(np.float32), 'hT': p['final_vector'].astype(np.float32),
     'farnn': 2, 'nl': fo.NL_CODES['tanh'], 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
gates = {'Wss1': rng.randn(S, S) * 0.1, 'Wrs1': rng.randn(R, S) * 0.1, 'bs1': np.full(S, 0.3),
         'Wss2': rng.randn(S, S) * 0.1, 'Wrs2': rng.randn(R, S) * 0.1, 'bs2': np.full(S, 0.2)}
gates = {k: v.astype(np.float32) for k, v in gates.items()}
q.update(gates)
h2 = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=2,
                             gates=gates, sigmoid_exponent=5, nl='tanh', o_idx=2)
worst = 0.0
for it in range(iters):
    B = int(rng.choice([1, 2, 9, 40, 130])); L = int(rng.choice([1, 6, 25]))
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    xd, ld = t(x), t(lengths)
    h2.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, None, scores.data_ptr())
    torch.cuda.synchronize()
    Lmax = int(lengths.max())
    ref = fo.decomp_ifst_scores(q, x, lengths)
    mask = np.arange(Lmax)[None, :] < lengths[:, None]
    err = np.abs(scores.cpu().numpy()[:, :Lmax][mask] - ref[mask]).max()
    worst = max(worst, float(err))
print('decomp: {} iterations, worst |score error| {:.2e}'.format(iters, worst))
sys.exit(1 if (bad or worst > 1e-4) else 0)

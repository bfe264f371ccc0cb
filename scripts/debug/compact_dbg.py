import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from re2nn_seq_amd import _lib, synth
rng = np.random.RandomState(0)
V, S, C, B, L = 7, 3, 3, 2, 4
T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=3.0, n_final=2)
x, lengths = synth.random_batch(V, B, L, rng, min_len=2)
print('h0', h0, 'hT', hT, 'o', O.sum(0), 'x', x, lengths)
for nl in ('none', 'relu', 'tanh', 'relutanh'):
    h = _lib.create_onehot_ifst(T, W, O, h0, hT, nl=nl)
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
    for compact in (False, True):
        h.set_compact(compact)
        sc = torch.empty((B, L, C), dtype=torch.float32, device='cuda')
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_FULL, None, None, sc.data_ptr())
        torch.cuda.synchronize()
        print(nl, 'compact' if compact else 'dense  ', sc.cpu().numpy()[0].round(4).tolist())

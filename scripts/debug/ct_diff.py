#!/usr/bin/env python3
"""compact_tag_kernel against the two-launch compact form and the oracle on the parity test's shapes: where the tags differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from re2nn_seq_amd import _lib, synth  # noqa: E402

for (S, C, L, B), nl in [((33, 5, 8, 3), 'none'), ((64, 9, 17, 5), 'none'), ((5, 3, 1, 4), 'tanh'), ((71, 128, 64, 40), 'none'), ((65, 130, 9, 3), 'relu')]:
    rng = np.random.RandomState(S * 31 + C)
    V = 29
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=max(2.0, S / 4), n_final=2)
    if '--no-overlap' not in sys.argv:
        W[0, min(1, S - 1)] = 1.0
        T[3 % V, 0, min(1, S - 1)] = 1.0
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    h = _lib.create_onehot_ifst(T, W, O, h0, hT, nl=nl, o_idx=1 % C)
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
    h.set_compact(True)
    for mode in (_lib.MODE_LOCAL, _lib.MODE_FULL):
        scores = torch.empty((B, L, C), dtype=torch.float32, device='cuda')
        tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, tags.data_ptr(), None, scores.data_ptr())
        torch.cuda.synchronize()
        outs = []
        for rep in range(3):
            tags2 = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
            h.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, tags2.data_ptr(), None, None)
            torch.cuda.synchronize()
            outs.append(tags2.cpu().numpy())
        t1 = tags.cpu().numpy()
        bad = np.argwhere(outs[0] != t1)
        print('S %d C %d L %d B %d %s mode %d: kernel %s; %d of %d differ; runs agree: %s' % (
            S, C, L, B, nl, mode, h.kernel_name(_lib.KERN_CHAIN), len(bad), B * L, all(np.array_equal(o, outs[0]) for o in outs)))
        for b, i in bad[:12]:
            print('   seq %d len %d pos %d: one launch %d, two launches %d%s' % (b, lengths[b], i, outs[0][b, i], t1[b, i],
                  '  (token %d)' % x[b, i]))
    h.close()

"""r04 debug: the shape-test failure S=125 C=129 L=129 B=2 relutanh max LOCAL scores=True (flat tags differ, tags equal)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import farnn_oracle as fo
from re2nn_seq_amd import _lib, synth
for seed in range(12):
    for S, C, L, B in ((125, 129, 129, 2), (71, 129, 129, 2), (104, 128, 129, 3)):
        rng = np.random.RandomState(seed)
        V = 37
        T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=max(2.0, S / 5), n_final=min(2, S))
        T = (T * 0.6).astype(np.float32)
        x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
        h = _lib.create_onehot_ifst(T, W, O, h0, hT, nl='relutanh', semiring='max', threshold=0.5, o_idx=1)
        xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
        for want_scores in (True, False):
            tags = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
            flat = torch.full((int(lengths.sum()),), -7, dtype=torch.int64, device='cuda')
            scores = torch.full((B, L, C), -7.0, dtype=torch.float32, device='cuda') if want_scores else None
            h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), flat.data_ptr(), scores.data_ptr() if want_scores else None)
            torch.cuda.synchronize()
            tg, fl = tags.cpu().numpy(), flat.cpu().numpy()
            mask = np.arange(L)[None, :] < lengths[:, None]
            bad = np.nonzero(tg[mask] != fl)[0]
            print(seed, S, 'scores' if want_scores else 'tags  ', h.kernel_name(_lib.KERN_CHAIN)[:40], 'lengths', lengths.tolist(), 'flat != tags at', bad[:10].tolist(), 'n', len(bad),
                  'flat vals', fl[bad[:6]].tolist(), 'tag vals', tg[mask][bad[:6]].tolist())
        h.close()

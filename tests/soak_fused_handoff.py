"""Soak of the fused chain + decode launch's inter-workgroup hand-off (csrc/chain.hip.h): bursts of back-to-back launches
with NO synchronisation in between, alternating between different batches on the same handle (so that a stale stash line
of the previous launch, at the same address, would decode into the previous batch's tags), two handles on two streams
so that workgroups of different launches share CUs (uneven load), every tag of every launch checked against the C port of
the oracle.  MI355X_MICROARCH.md: "test every hand-off under UNEVEN load, consumer L1-warm, checking every word".

    python tests/soak_fused_handoff.py [rounds]        (also run, short, as tests/test_gpu_parity_onehot.py::test_fused_handoff_burst)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_port                                # noqa: E402
from re2nn_seq_amd import _lib, synth                    # noqa: E402


def run(rounds=20, burst=48, seed=0, verbose=True):
    rng = np.random.RandomState(seed)
    V, S, C = 950, 71, 129
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
    Tf = (T + W).astype(np.float32)
    handles = [_lib.create_onehot_ifst(T, W, O, h0, hT, o_idx=0) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = 0
    for r in range(rounds):
        B = int(rng.choice([64, 200, 256, 300]))
        L = int(rng.choice([33, 64, 100]))
        nb = 4
        batches = []
        for _ in range(nb):
            x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
            if rng.rand() < 0.3:
                lengths[:] = L
            ref, _, _ = c_port.onehot_ifst_tag(Tf, O, h0, hT, x, lengths, threshold=0.5, o_idx=0, nthreads=8)
            batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda(), ref, lengths))
        outs = [torch.full((B, L), -7, dtype=torch.int32, device='cuda') for _ in range(burst)]
        for hh in handles:
            hh.reserve(B, L)
        torch.cuda.synchronize()
        for i in range(burst):                            # no sync inside the burst; consecutive launches differ in batch
            k = i % 2
            xd, ld, _, _ = batches[(i // 2 + i) % nb]
            handles[k].tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, outs[i].data_ptr(), None, None,
                           streams[k].cuda_stream)
        torch.cuda.synchronize()
        for i in range(burst):
            _, _, ref, lengths = batches[(i // 2 + i) % nb]
            mask = np.arange(L)[None, :] < lengths[:, None]
            got = outs[i].cpu().numpy()
            if not (np.array_equal(got[mask], ref[mask]) and (got[~mask] == -1).all()):
                bad += 1
                if verbose:
                    print('MISMATCH round', r, 'launch', i, 'B', B, 'L', L, 'wrong tags', int((got[mask] != ref[mask]).sum()))
    for hh in handles:
        hh.close()
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad = run(rounds=n)
    print('soak: {} rounds x 48 launches, {} launches with a mismatch'.format(n, bad))
    sys.exit(1 if bad else 0)

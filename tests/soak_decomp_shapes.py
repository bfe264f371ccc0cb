"""Random geometries of the decomposed i-FST (states, rank, gates, CRF, batch, length) through farnn_tag, scores against the
oracle (1e-4): covers the register kernel, the rows kernel's register forms (upper bounds on passes / chunks), the LDS +
L2 path, the score tiles and the Viterbi layouts at shapes no fixed test names.    python tests/soak_decomp_shapes.py [n]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import farnn_oracle as fo                    # noqa: E402
from re2nn_seq_amd import _lib, synth                    # noqa: E402


def run(n=60, seed=0, verbose=True, only=None):
    """only: replay the random stream but run just that iteration (a mismatch's geometry, e.g. under other FARNN_* switches)"""
    rng = np.random.RandomState(seed)
    f = lambda a: np.asarray(a, np.float32)              # noqa: E731
    bad = sensitive = 0
    for it in range(n):
        S = int(rng.choice([5, 17, 40, 64, 71, 96, 104, 128, 134, 150]))
        R = int(rng.choice([3, 20, 50, 64, 65, 100, 150, 250]))
        farnn = int(rng.randint(0, 3))
        crf = bool(rng.rand() < 0.3)
        C = int(rng.choice([5, 30, 73, 126]))
        V, B, L = 150, int(rng.choice([1, 7, 24, 40])), int(rng.choice([3, 17, 33, 64]))
        p = synth.random_decomposed_params(V, S, C, R, 20, rng, contractive=True)
        Cout = f(p['C_output_mat'])
        tr = None
        if crf:
            Cout = np.concatenate([Cout, (rng.rand(2, S) * 0.01).astype(np.float32)], 0)
            tr = fo.crf_default_transitions(C) + (rng.randn(C + 2, C + 2) * 1.0).astype(np.float32)
        q = {'Vgen': f(p['V_embed']), 'S1': f(p['S1']), 'S2': f(p['S2']), 'W': f(p['wildcard_mat']), 'Cout': Cout,
             'h0': f(p['start_vector']), 'hT': f(p['final_vector']), 'farnn': farnn, 'nl': fo.NL_TANH,
             'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
        gates = None
        if farnn:
            gates = {'Wss1': f(rng.randn(S, S) * 0.03), 'Wrs1': f(rng.randn(R, S) * 0.03), 'bs1': f(np.full(S, 1.0))}
            if farnn == 2:
                gates.update(Wss2=f(rng.randn(S, S) * 0.03), Wrs2=f(rng.randn(R, S) * 0.03), bs2=f(np.full(S, 1.0)))
            q.update(gates)
        x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
        if only is not None and it != only:
            continue
        h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn, gates=gates,
                                    sigmoid_exponent=5, nl='tanh', threshold=0.5, o_idx=0, use_crf=crf, crf_trans=tr)
        K = Cout.shape[0]
        xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
        scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
        tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
        tags2 = torch.empty((B, L), dtype=torch.int32, device='cuda')
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, scores.data_ptr())
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags2.data_ptr(), None, None)
        torch.cuda.synchronize()
        ref = fo.decomp_ifst_scores(q, x, lengths)
        mask = np.arange(L)[None, :] < lengths[:, None]
        got = scores.cpu().numpy()
        # ONE rule (round 5; the round-4 soak excused a geometry whenever the float32 oracle was itself noisy and clamped its error
        # for the report): every geometry is held to the EXACT value -- the oracle evaluated in float64.  The kernel's distance from
        # it must stay within the 1e-4 bar, or, on a locally chaotic random gated model, within TWICE the distance the float32
        # oracle itself keeps from it (two float32 evaluations of such a recurrence are two draws from the same scatter).
        # Geometries that pass only through the second clause are counted and listed with their unclamped numbers; more than
        # 0.5 % of them fails the run.
        with fo.precision(np.float64):
            ref64 = fo.decomp_ifst_scores({k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == 'f' else v)
                                           for k, v in q.items()}, x, lengths)
        err = np.abs(got[mask] - ref[mask]).max() if mask.any() else 0.0
        e64 = np.abs(got[mask] - ref64[mask]).max() if mask.any() else 0.0
        noise = np.abs(ref[mask] - ref64[mask]).max() if mask.any() else 0.0
        bar = 1e-4 + 1e-4 * np.abs(ref64[mask]).max() if mask.any() else 1e-4
        note = ''
        if e64 > bar or only is not None:
            note = ' [kernel vs float64 {:.2e}, float32 oracle vs float64 {:.2e}, kernel vs float32 oracle {:.2e}, bar {:.2e}]'.format(e64, noise, err, bar)
        if e64 > bar and e64 <= 2 * noise:
            sensitive += 1
            note += ' SENSITIVE MODEL: within twice the float32 oracle\'s own distance from the exact value'
        ok = np.isfinite(got).all() and e64 <= max(bar, 2 * noise)
        if crf:     # Viterbi on the GPU's own scores, fused and unfused launches
            own = fo.decode_crf(got, lengths, tr, 0.5, 0)
            ok = ok and np.array_equal(own[mask], tags.cpu().numpy()[mask]) and np.array_equal(own[mask], tags2.cpu().numpy()[mask])
        else:
            ok = ok and np.array_equal(tags.cpu().numpy()[mask], tags2.cpu().numpy()[mask])
        if not ok:
            bad += 1
        if verbose and (not ok or note or it % 10 == 0):
            print('{} S={} R={} farnn={} crf={} C={} B={} L={} kernel={} err={:.2e} {}{}'.format(
                it, S, R, farnn, crf, C, B, L, h.kernel_name(_lib.KERN_CHAIN), err, 'ok' if ok else 'MISMATCH', note), flush=True)
        h.close()
    if sensitive:
        print('({} of {} geometries beyond the 1e-4 bar but within twice the float32 oracle\'s own distance from float64: listed above)'.format(sensitive, n))
    if sensitive > max(2, n // 200):
        print('too many of them: counted as a failure')
        bad += 1
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = run(n, only=int(sys.argv[2]) if len(sys.argv) > 2 else None)
    print('soak: {} random decomposed geometries, {} mismatches'.format(n, bad))
    sys.exit(1 if bad else 0)

"""GPU parity AT THE SIZES bench.py times (BASELINE.json configs[2] and configs[3]), through the C-ABI:
the batch geometry matters -- two-round / paired scheduling of the decomposed recurrence over 512 chains,
the in-kernel length-rank selection at B = 256, the 11 000-row word-table gather, the K = 73 / K = 130 score
tiles -- so the small-geometry tests do not stand in for these.

    config 2   SNIPS-BIO-sized decomposed i-FST: V=11 000, S=104, C=73, B=256, L=64, lengths U[5,64]
               (i)  rank 50, farnn 0, tanh                (what `bench.py --workload decomp` times)
               (ii) rank 250, farnn 2, use_crf=1          (the shape of the shipped example `.res` configurations)
    config 3   ATIS-BIO-sized onehot i-FST + fused Viterbi: V=950, S=71, C=128 (+2), B=256, L=64

Oracle: oracle/farnn_oracle.py (pinned to the reference by tests/test_oracle_golden.py).  Bar: scores within
1e-4 (north_star), tags equal wherever the oracle's decision margin exceeds 2e-4 (twice the score bar); bit-exact for the onehot path.
Reference: model_decompose_single.py:207-304, model_decompose.py:339-371, crf.py:102-195."""
import numpy as np
import pytest
import torch

from oracle import farnn_oracle as fo

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _snips_model(R, farnn, crf, seed=1234, S=104):
    """The generator bench.py uses for `--workload decomp` (same seed), plus gates / CRF rows on demand."""
    from re2nn_seq_amd import synth
    V, C = 11000, 73
    wrng = np.random.RandomState(seed)
    p = synth.random_decomposed_params(V, S, C, R, 100, wrng, contractive=True)
    f = lambda a: np.asarray(a, np.float32)                       # noqa: E731
    Cout = f(p['C_output_mat'])
    tr = None
    if crf:            # two extra rows for START / STOP (model_decompose_single.py:78-79), small random values
        Cout = np.concatenate([Cout, (wrng.rand(2, S) * 0.01).astype(np.float32)], 0)
        K = C + 2
        tr = fo.crf_default_transitions(C) + (wrng.randn(K, K) * 1.0).astype(np.float32)
    q = {'Vgen': f(p['V_embed']), 'S1': f(p['S1']), 'S2': f(p['S2']), 'W': f(p['wildcard_mat']), 'Cout': Cout,
         'h0': f(p['start_vector']), 'hT': f(p['final_vector']), 'farnn': farnn, 'nl': fo.NL_TANH,
         'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
    gates = None
    if farnn:
        gates = {'Wss1': f(wrng.randn(S, S) * 0.03), 'Wrs1': f(wrng.randn(R, S) * 0.03), 'bs1': f(np.full(S, 1.0))}
        if farnn == 2:
            gates.update(Wss2=f(wrng.randn(S, S) * 0.03), Wrs2=f(wrng.randn(R, S) * 0.03), bs2=f(np.full(S, 1.0)))
        q.update(gates)
    return V, q, gates, tr


@pytest.mark.parametrize('R,farnn,crf,S', [(50, 0, False, 104), (250, 2, True, 104), (100, 1, False, 104), (100, 2, False, 104),
                                            (150, 2, True, 104), (150, 2, False, 134), (150, 1, False, 134), (250, 2, False, 134),
                                            (150, 2, True, 134), (200, 2, False, 150), (120, 0, False, 71)])
def test_decomposed_ifst_at_bench_size_vs_oracle(R, farnn, crf, S):
    """(the S = 134 / 150 cases: `--additional_states 30` on top of the automaton, as two shipped example configurations have it
    -- the rows kernel's MIXED register forms: first passes of the gate / output rows in registers, the rows behind them in LDS;
    the ungated rank-120 case: the rows kernel's P2-in-registers form)"""
    from re2nn_seq_amd import _lib, synth
    B, L = 256, 64
    V, q, gates, tr = _snips_model(R, farnn, crf, S=S)
    x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))       # bench.py's rank-0 batch
    assert lengths.min() >= 5 and lengths.max() == L
    K = q['Cout'].shape[0]
    h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn,
                                gates=gates, sigmoid_exponent=5, nl='tanh', threshold=0.5, o_idx=0,
                                use_crf=crf, crf_trans=tr)
    assert h.kernel_name(_lib.KERN_CHAIN) in ('decomp_rows_kernel', 'decomp_regs_kernel')
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
    flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), flat.data_ptr(), scores.data_ptr())
    # the call bench.py times: tags only (no score tensor: the fused score+Viterbi kernel when CRF is on)
    tags2 = torch.empty((B, L), dtype=torch.int32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags2.data_ptr(), None, None)
    torch.cuda.synchronize()

    ref = fo.decomp_ifst_scores(q, x, lengths)                    # [B, L, K]
    mask = np.arange(L)[None, :] < lengths[:, None]
    got = scores.cpu().numpy()
    if S >= 134:
        # the rank-250 model of this size has one locally sensitive sequence (b = 77): numpy's own float32 evaluations of it at batch 1 and at batch
        # 256 differ by 2e-4, the float64 evaluation lies between them.  Held to the float64 value, which every float32
        # implementation scatters around (the kernels sit 6e-5 from it); the float32 oracle to the bar its own noise allows
        with fo.precision(np.float64):
            ref64 = fo.decomp_ifst_scores({k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == 'f' else v)
                                           for k, v in q.items()}, x, lengths)
        np.testing.assert_allclose(got[mask], ref64[mask], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=4e-4)
        assert (np.abs(got - ref)[mask] > 1e-4 + 1e-4 * np.abs(ref[mask])).mean() < 1e-4
    else:
        np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4)
    assert (got[~mask] == 0).all()
    tg, tg2 = tags.cpu().numpy(), tags2.cpu().numpy()
    assert (tg[~mask] == -1).all() and (tg2[~mask] == -1).all()
    if not crf:
        rt = fo.decode_argmax(ref, 0.5, 0)
        refc = ref.copy(); refc[..., -1] = np.minimum(refc[..., -1], 0.5)
        top2 = np.sort(refc[mask], axis=1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 2e-4                    # twice the 1e-4 score bar: what that bar justifies
        assert safe.mean() > 0.9                                   # the comparison is not vacuous
        assert np.array_equal(tg[mask][safe], rt[mask][safe])
        assert np.array_equal(tg2[mask][safe], rt[mask][safe])
        assert np.array_equal(flat.cpu().numpy()[safe], fo.forward_local_tags(ref, lengths, 0.5, 0)[safe])
    else:
        # Viterbi on the GPU's own scores must equal the oracle's Viterbi on those scores bit for bit (the DP is
        # the same f32 expression); against the oracle's scores a path may differ only where the two totals tie
        # within float noise, so hold whole sequences equal and allow a handful of near-tie sequences
        own = fo.decode_crf(got, lengths, tr, 0.5, 0)
        assert np.array_equal(own[mask], tg[mask].astype(np.int64))
        assert np.array_equal(own[mask], tg2[mask].astype(np.int64))       # fused kernel == score kernel + Viterbi
        assert np.array_equal(fo.flatten(own, lengths), flat.cpu().numpy())
        # ... and EVERY sequence whose path differs from the oracle's path must be as good a path as the oracle's under the
        # ORACLE's scores: |path score - best path score| <= 1e-3 * |score| (a tie within the float noise of 1e-4 scores
        # summed over <= 64 positions), not "97 % of the sequences equal"
        sref = np.array(ref, dtype=np.float32, copy=True)
        sref[..., K - 3] = np.minimum(sref[..., K - 3], np.float32(0.5))
        want_raw = fo.viterbi_paths(sref, lengths, tr)
        START, STOP = K - 2, K - 1

        def path_score(sc, path):
            t = float(sc[0, path[0]] + tr[START, path[0]])
            for i in range(1, len(path)):
                t += float(sc[i, path[i]] + tr[path[i - 1], path[i]])
            return t + float(tr[path[-1], STOP])

        n_diff = 0
        for b in range(B):
            n = int(lengths[b])
            got_b, want_b = tg[b, :n].astype(np.int64), want_raw[b, :n].astype(np.int64)
            want_mapped = np.where(want_b == K - 3, 0, want_b)
            if np.array_equal(got_b, want_mapped):
                continue
            n_diff += 1
            # undo the K-3 -> o_idx mapping of the GPU's tags: where the oracle's raw tag maps to the same id, take it
            raw = got_b.copy()
            amb = (got_b == 0)
            raw[amb & (want_b == K - 3)] = K - 3
            best = path_score(sref[b], want_b)
            mine = path_score(sref[b], raw)
            for i in np.nonzero(amb & (want_b != K - 3) & (want_b != 0))[0]:      # ambiguous inside a differing stretch: the better reading
                alt = raw.copy(); alt[i] = K - 3
                mine = max(mine, path_score(sref[b], alt))
            assert mine <= best + 1e-3 * abs(best) and mine >= best - 1e-3 * abs(best), (b, mine, best)
        assert n_diff <= B // 8, n_diff                            # (near-ties are rare; a flood of them is a bug)
    h.close()


def test_onehot_ifst_crf_at_bench_size_vs_oracle():
    """BASELINE configs[3] exactly as `bench.py --workload ifst_crf` builds it: V=950, S=71, C=128, K=130,
    B=256, L=64, random transitions.  Integer scores: the decoded paths are bit-identical."""
    from oracle import c_port
    from re2nn_seq_amd import _lib, synth
    V, S, C, B, L = 950, 71, 128, 256, 64
    wrng = np.random.RandomState(1234)
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, wrng)
    K = C + 2
    tr = (wrng.randn(K, K) * 0.1).astype(np.float32)
    tr[:, K - 2] = -10000.0
    tr[K - 1, :] = -10000.0
    x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))
    h = _lib.create_onehot_ifst(T, W, O, h0, hT, use_crf=True, crf_trans=tr)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
    flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), flat.data_ptr(), None)   # fused
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    tags_u = torch.empty((B, L), dtype=torch.int32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_u.data_ptr(), None, scores.data_ptr())  # unfused
    torch.cuda.synchronize()
    _, sc, _ = c_port.onehot_ifst_tag(T + W, O, h0, hT, x, lengths, want_scores=True, nthreads=8)
    ext = fo.onehot_crf_extension_scores(sc)
    mask = np.arange(L)[None, :] < lengths[:, None]
    assert np.array_equal(scores.cpu().numpy()[mask], ext[mask])
    want = fo.decode_crf(ext, lengths, tr, 0.5, 0)
    assert np.array_equal(want[mask], tags.cpu().numpy().astype(np.int64)[mask])
    assert np.array_equal(want[mask], tags_u.cpu().numpy().astype(np.int64)[mask])
    assert np.array_equal(fo.flatten(want, lengths), flat.cpu().numpy())
    assert (tags.cpu().numpy()[~mask] == -1).all()
    h.close()


def test_register_forms_do_not_see_what_earlier_kernels_left_in_lds():
    """The rows kernel's register forms read whole (upper-bound) chunk counts of their input vectors and run on into the LDS
    behind them with zero weights; LDS keeps what earlier workgroups wrote there -- the Viterbi kernel's -inf pads would
    turn 0 x garbage into NaN.  A Viterbi launch that covers every CU, then a gated decomposed model: scores finite and
    equal to the oracle's."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(3)
    Vc, Sc, Cc = 300, 71, 128
    T, W, O, h0, hT = synth.random_ifst_tensors(Vc, Sc, Cc, rng, edges_per_word=8.0)
    trc = fo.crf_default_transitions(Cc) + rng.randn(Cc + 2, Cc + 2).astype(np.float32)
    hc = _lib.create_onehot_ifst(T, W, O, h0, hT, o_idx=0, use_crf=True, crf_trans=trc)
    xc, lc = synth.random_batch(Vc, 512, 64, rng, min_len=40)
    xcd, lcd = _t(xc).cuda(), _t(lc).cuda()
    tc = torch.empty((512, 64), dtype=torch.int32, device='cuda')
    B, L = 256, 64
    V, q, gates, tr = _snips_model(100, 2, False)
    x, lengths = synth.random_batch(V, B, L, np.random.RandomState(4321))
    h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=2, gates=gates,
                                sigmoid_exponent=5, nl='tanh', threshold=0.5, o_idx=0)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    K = q['Cout'].shape[0]
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    ref = fo.decomp_ifst_scores(q, x, lengths)
    mask = np.arange(L)[None, :] < lengths[:, None]
    for _ in range(3):
        hc.tag(xcd.data_ptr(), lcd.data_ptr(), 512, 64, _lib.MODE_LOCAL, tc.data_ptr(), None, None)
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, None, scores.data_ptr())
        torch.cuda.synchronize()
        got = scores.cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4)
    h.close(); hc.close()

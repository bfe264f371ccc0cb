"""GPU parity of the decomposed i-FST path (FARNN_S_D_W_I_S mirror -> C-ABI -> HIP) and of the
fused Viterbi decode, against the reference fixtures and the oracle.  Float path: scores within
1e-4 (north_star); tags equal."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import farnn_oracle as fo
from util import ns, load_golden, GOLDEN, assert_float_path, in_float64

pytestmark = pytest.mark.gpu


def _configs():
    with open(os.path.join(GOLDEN, 'decomp_small.json')) as f:
        return json.load(f)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _build(g, k, cfg, meta):
    from re2nn_seq_amd.farnn.model_decompose_single import FARNN_S_D_W_I_S
    a = ns(independent=2, threshold=meta['threshold'], **cfg)
    torch.manual_seed(0)
    m = FARNN_S_D_W_I_S(V=g['V_in'], S1=g['S1_in'], S2=g['S2_in'], C_output_mat=g['O_in'],
                        wildcard_mat=g['W_in'], wildcard_output_vector=g['Ow_in'],
                        final_vector=g['final_in'], start_vector=g['start_in'],
                        pretrained_word_embed=g['E_in'], priority_mat=g['priority_in'], args=a,
                        o_idx=meta['o_idx'])
    # exact parameters of the captured reference model (random pads / gates / CRF transitions)
    pre = 'c{}.'.format(k)
    sd = {}
    for key in g.files:
        if key.startswith(pre) and key not in (pre + 'scores', pre + 'flat_pred'):
            name = key[len(pre):]
            name = {'embedding': 'embedding.weight', 'priority_mat': 'priority_layer.priority_mat',
                    'crf_transitions': 'crf.transitions'}.get(name, name)
            sd[name] = g[key]
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize('k', range(len(_configs()['configs'])))
def test_decomposed_small_vs_reference(k):
    from re2nn_seq_amd import _lib
    meta = _configs()
    cfg = meta['configs'][k]
    g = load_golden('decomp_small')
    x, lengths = g['x'], g['lengths']
    m = _build(g, k, cfg, meta)
    pre = 'c{}.'.format(k)
    ref_scores = g[pre + 'scores']
    Lmax = int(lengths.max())
    assert ref_scores.shape[1] == Lmax
    # FULL mode over the first Lmax columns = the reference's padded loop (ref :221, :236-249)
    r = m.run(_t(x[:, :Lmax]), _t(lengths), _lib.MODE_FULL, want_scores=True)
    np.testing.assert_allclose(r['scores'].cpu().numpy(), ref_scores, rtol=1e-4, atol=1e-4)
    _, pred, true = m.forward_local(_t(x), torch.zeros_like(_t(x)), _t(lengths), train=False)
    assert pred.dtype == torch.int64
    assert np.array_equal(pred.numpy(), g[pre + 'flat_pred'])
    assert true.shape == pred.shape


def test_decomposed_constructor_matches_reference_shapes():
    """The mirror's own constructor (no state loading) builds the reference's parameter set."""
    meta = _configs()
    g = load_golden('decomp_small')
    for k, cfg in enumerate(meta['configs']):
        m = _build(g, k, cfg, meta)
        a = ns(independent=2, **cfg)
        S = g['S1_in'].shape[0] + a.additional_states
        K = g['O_in'].shape[0] + (2 if a.use_crf else 0)
        assert tuple(m.S1.shape) == (S, g['S1_in'].shape[1])
        assert tuple(m.C_output_mat.shape) == (K, S)
        assert tuple(m.wildcard_mat.shape) == (S, S)
        assert m.handle.num_columns() == K


@pytest.mark.parametrize('tr_kind', ['default', 'random'])
def test_onehot_ifst_with_fused_viterbi(tr_kind):
    """BASELINE config 4 (onehot + use_crf=1): the composition SURVEY.md 8a-note defines, checked
    against the oracle's restatement of the reference's CRF decode chain."""
    from re2nn_seq_amd.farnn.model_onehot import FARNN_S_O_I_S
    g = load_golden('ifst_small')
    S, C = g['T'].shape[1], g['O'].shape[0]
    x, lengths, o_idx = g['x'], g['lengths'], int(g['o_idx'])
    rng = np.random.RandomState(17)
    tr = fo.crf_default_transitions(C)
    if tr_kind == 'random':
        tr = tr + rng.randn(C + 2, C + 2).astype(np.float32) * 0.5
    m = FARNN_S_O_I_S(g['T'], g['O'], g['W'], np.zeros(S), g['hT'], g['h0'], None, ns(), o_idx=o_idx)
    m.enable_crf(tr)
    _, pred, _ = m.forward_local(_t(x), torch.zeros_like(_t(x)), _t(lengths), train=False)
    sc = fo.onehot_crf_extension_scores(fo.onehot_ifst_scores(g['T'], g['W'], g['O'], g['h0'], g['hT'], x, lengths))
    ref = fo.forward_local_tags(sc, lengths, 0.5, o_idx, crf_tr=tr)
    assert np.array_equal(pred.numpy(), ref)


@pytest.mark.parametrize('variant', ['fused', 'history', 'backpointers'])
def test_viterbi_atis_scale_vs_oracle(variant, monkeypatch):
    """K=130 tags, L=64, B=64: the Viterbi kernels -- scores computed inside the decode kernel (fused),
    partition history + lazy back-pointers behind the separate score kernel, and the stored-back-pointer
    fallback for tag sets whose history does not fit the LDS."""
    monkeypatch.setenv('FARNN_VITERBI_BP', '1' if variant == 'backpointers' else '0')
    monkeypatch.setenv('FARNN_VITERBI_UNFUSED', '0' if variant == 'fused' else '1')
    from re2nn_seq_amd import synth
    from re2nn_seq_amd.farnn.model_onehot import FARNN_S_O_I_S
    rng = np.random.RandomState(5)
    V, S, C, B, L = 300, 71, 128, 64, 64
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=8.0)
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    tr = fo.crf_default_transitions(C) + rng.randn(C + 2, C + 2).astype(np.float32)
    m = FARNN_S_O_I_S(T, O, W, np.zeros(S), hT, h0, None, ns(), o_idx=3).enable_crf(tr)
    _, pred, _ = m.forward_local(_t(x), torch.zeros_like(_t(x)), _t(lengths), train=False)
    sc = fo.onehot_crf_extension_scores(fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths))
    assert np.array_equal(pred.numpy(), fo.forward_local_tags(sc, lengths, 0.5, 3, crf_tr=tr))


@pytest.mark.parametrize('variant', ['fused', 'history'])
@pytest.mark.parametrize('C', [2, 3, 6, 13, 28, 29, 30, 31, 45, 62, 73, 94, 95, 98, 110, 125, 126, 127, 128, 129, 135, 158, 198, 253, 254])
def test_viterbi_tag_set_sizes_vs_oracle(C, variant, monkeypatch):
    """Every shape of the Viterbi kernel's work split (csrc/score_decode.hip.h: eight lanes per tag pair, 32-source blocks +
    0..4 leftover slots, the tail wavefront at 64 / 32 / 16 / 8 lanes per pair, K = C + 2 from 4 to 256), scores
    computed inside the kernel and read from the score kernel's output, ragged lengths incl. 1: tags equal the oracle's
    (crf.py:102-195) bit for bit."""
    monkeypatch.setenv('FARNN_VITERBI_BP', '0')
    monkeypatch.setenv('FARNN_VITERBI_UNFUSED', '0' if variant == 'fused' else '1')
    from re2nn_seq_amd import synth
    from re2nn_seq_amd.farnn.model_onehot import FARNN_S_O_I_S
    rng = np.random.RandomState(100 + C)
    V, S, B, L = 60, 19 + C % 7, 9, 5 + C % 13
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=8.0)
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    lengths[0] = 1
    o_idx = C // 2
    tr = fo.crf_default_transitions(C) + rng.randn(C + 2, C + 2).astype(np.float32)
    m = FARNN_S_O_I_S(T, O, W, np.zeros(S), hT, h0, None, ns(), o_idx=o_idx).enable_crf(tr)
    _, pred, _ = m.forward_local(_t(x), torch.zeros_like(_t(x)), _t(lengths), train=False)
    sc = fo.onehot_crf_extension_scores(fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths))
    assert np.array_equal(pred.numpy(), fo.forward_local_tags(sc, lengths, 0.5, o_idx, crf_tr=tr))


# ---------------------------------------------------------------- decomposed independent=1 (a15)
def _ind1_configs():
    with open(os.path.join(GOLDEN, 'decomp_ind1_small.json')) as f:
        return json.load(f)


def _build_ind1(g, k, cfg, meta):
    from re2nn_seq_amd.farnn.model_decompose_independent import FARNN_S_D_W_I
    a = ns(independent=1, threshold=meta['threshold'], **cfg)
    torch.manual_seed(0)
    m = FARNN_S_D_W_I(V=g['V_in'], S1=g['S1_in'], S2=g['S2_in'], C_output=g['C_in'],
                      S1_output=g['S1o_in'], S2_output=g['S2o_in'], wildcard_mat=g['W_in'],
                      wildcard_output=None, final_vector=g['final_in'], start_vector=g['start_in'],
                      pretrained_word_embed=g['E_in'], priority_mat=g['priority_in'], args=a,
                      o_idx=meta['o_idx'])
    pre = 'c{}.'.format(k)
    sd = {}
    for key in g.files:
        if key.startswith(pre) and key not in (pre + 'scores', pre + 'flat_pred'):
            name = key[len(pre):]
            name = {'embedding': 'embedding.weight', 'priority_mat': 'priority_layer.priority_mat',
                    'crf_transitions': 'crf.transitions'}.get(name, name)
            sd[name] = g[key]
    shapes = {n: tuple(getattr(m, n).shape) for n in ('S1', 'S2', 'C_output', 'S1_output', 'S2_output',
                                                      'wildcard_mat', 'h0', 'hT')}
    m.load_state_dict(sd)
    for n, shp in shapes.items():      # the mirror's constructor built the reference's shapes
        assert tuple(getattr(m, n).shape) == shp, n
    return m


@pytest.mark.parametrize('k', range(len(_ind1_configs()['configs'])))
def test_decomposed_independent1_vs_reference(k):
    """FARNN_S_D_W_I (--independent 1): scores within 1e-4 of the captured reference, tags equal."""
    from re2nn_seq_amd import _lib
    meta = _ind1_configs()
    cfg = meta['configs'][k]
    g = load_golden('decomp_ind1_small')
    x, lengths = g['x'], g['lengths']
    m = _build_ind1(g, k, cfg, meta)
    pre = 'c{}.'.format(k)
    ref_scores = g[pre + 'scores']
    Lmax = int(lengths.max())
    r = m.run(_t(x[:, :Lmax]), _t(lengths), _lib.MODE_FULL, want_scores=True)
    np.testing.assert_allclose(r['scores'].cpu().numpy(), ref_scores, rtol=1e-4, atol=1e-4)
    _, pred, true = m.forward_local(_t(x), torch.zeros_like(_t(x)), _t(lengths), train=False)
    assert np.array_equal(pred.numpy(), g[pre + 'flat_pred'])
    assert true.shape == pred.shape


def test_decomposed_independent1_ragged_vs_oracle():
    """A larger ragged batch (B=48, L=24, S=23, R=40, RO=30), LOCAL mode, against the oracle."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(11)
    V, S, R, RO, K, B, L = 90, 23, 40, 30, 9, 48, 24
    p = {'Vgen': (rng.randn(V, R) * 0.4).astype(np.float32), 'S1': (rng.randn(S, R) * 0.4).astype(np.float32),
         'S2': (rng.randn(S, R) * 0.4).astype(np.float32), 'W': (rng.rand(S, S) < 0.1).astype(np.float32) * 0.5,
         'Cout': (rng.randn(K, RO) * 0.5).astype(np.float32), 'S1o': (rng.randn(S, RO) * 0.2).astype(np.float32),
         'S2o': (rng.randn(S, RO) * 0.2).astype(np.float32), 'Wo': None,     # contractive recurrence
         'h0': np.eye(S, dtype=np.float32)[0], 'hT': (rng.rand(S) < 0.3).astype(np.float32),
         'farnn': 0, 'nl': fo.NL_CODES['tanh'], 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    h = _lib.create_decomp_ind1(p['Vgen'], p['S1'], p['S2'], p['W'], p['Cout'], p['S1o'], p['S2o'],
                                p['h0'], p['hT'], nl='tanh', threshold=0.5, o_idx=2)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
    flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), flat.data_ptr(),
          scores.data_ptr())
    torch.cuda.synchronize()
    Lmax = int(lengths.max())
    ref = fo.decomp_ind1_scores(p, x, lengths)
    got = scores.cpu().numpy()[:, :Lmax]
    mask = np.arange(Lmax)[None, :] < lengths[:, None]
    np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4)
    assert np.array_equal(flat.cpu().numpy(), fo.forward_local_tags(ref, lengths, 0.5, 2))
    t = tags.cpu().numpy()
    assert (t[:, :Lmax][~mask] == -1).all()


@pytest.mark.parametrize('S,RO,K,B,L,prio,crf', [
    (100, 70, 73, 24, 20, True, False),     # the bench geometry: 7 row tiles, 5 column tiles, 2 label columns per lane
    (112, 80, 130, 9, 12, False, False),    # largest shape of the MFMA path; 3 label-column passes
    (16, 16, 5, 5, 9, True, False),         # one tile of everything
    (17, 15, 7, 3, 6, False, True),         # ragged tiles, CRF emissions
    (113, 20, 9, 4, 7, True, False),        # S above the register budget: generic kernel
    (30, 81, 9, 4, 7, False, False),        # output rank above 80: generic kernel
    (40, 33, 70, 1, 33, True, False),       # one sequence
])
def test_decomposed_independent1_scoring_geometries_vs_oracle(S, RO, K, B, L, prio, crf):
    """The independent=1 scoring kernels (per-word table on the f32 MFMA + label kernel, and the generic
    fallback) over tile-boundary shapes, with empty sequences, against the oracle; FULL mode too."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(S * 7 + RO)
    V, R = 60, 12
    p = {'Vgen': (rng.randn(V, R) * 0.4).astype(np.float32), 'S1': (rng.randn(S, R) * 0.3).astype(np.float32),
         'S2': (rng.randn(S, R) * 0.3).astype(np.float32), 'W': (rng.rand(S, S) < 0.1).astype(np.float32) * 0.5,
         'Cout': (rng.randn(K, RO) * 0.5).astype(np.float32), 'S1o': (rng.randn(S, RO) * 0.1).astype(np.float32),
         'S2o': (rng.randn(S, RO) * 0.1).astype(np.float32), 'Wo': None,
         'h0': np.eye(S, dtype=np.float32)[0], 'hT': (rng.rand(S) < 0.3).astype(np.float32),
         'farnn': 0, 'nl': fo.NL_CODES['tanh'], 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
    P = (np.eye(K) + (rng.rand(K, K) < 0.05) * 0.5).astype(np.float32) if prio else None
    tr = None
    if crf:
        tr = fo.crf_default_transitions(K - 2) + (rng.randn(K, K) * 0.1).astype(np.float32)
        tr[:, K - 2] = -10000.0
        tr[K - 1, :] = -10000.0
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    if B > 3:
        lengths[1] = 0                                   # an empty sequence in the middle of the batch
        lengths[B - 1] = L
    h = _lib.create_decomp_ind1(p['Vgen'], p['S1'], p['S2'], p['W'], p['Cout'], p['S1o'], p['S2o'],
                                p['h0'], p['hT'], P=P, nl='tanh', threshold=0.5, o_idx=2, use_crf=crf,
                                crf_trans=tr)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    tags = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
    flat = torch.empty((max(int(lengths.sum()), 1),), dtype=torch.int64, device='cuda')
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), flat.data_ptr(),
          scores.data_ptr())
    torch.cuda.synchronize()
    Lmax = max(int(lengths.max()), 1)
    ref = fo.decomp_ind1_scores(p, x, lengths, P=P)
    got = scores.cpu().numpy()[:, :Lmax]
    mask = np.arange(Lmax)[None, :] < lengths[:, None]
    assert_float_path(got[mask], ref[mask], in_float64(fo.decomp_ind1_scores, p, x, lengths, P=P)[mask])      # the ONE 1e-4 rule (util.py)
    assert (scores.cpu().numpy()[~(np.arange(L)[None, :] < lengths[:, None])] == 0).all()
    want = fo.forward_local_tags(got, lengths, 0.5, 2, crf_tr=tr)     # decode the kernel's own scores: exact
    assert np.array_equal(flat.cpu().numpy()[:int(lengths.sum())], want)
    if not crf:
        t = tags.cpu().numpy()
        assert (t[~(np.arange(L)[None, :] < lengths[:, None])] == -1).all()
        assert np.array_equal(t[np.arange(L)[None, :] < lengths[:, None]], want)
    # the same call without flat output or scores (the flat offsets are still needed to slice the batch)
    tags2 = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags2.data_ptr(), None, None)
    torch.cuda.synchronize()
    if not crf:
        assert torch.equal(tags2, tags)
    # FULL mode: every position is scored; the valid ones must agree with LOCAL mode
    scores_f = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    tags_f = torch.empty((B, L), dtype=torch.int32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_FULL, tags_f.data_ptr(), None, scores_f.data_ptr())
    torch.cuda.synchronize()
    gf = scores_f.cpu().numpy()
    assert np.isfinite(gf).all()
    valid = np.arange(L)[None, :] < lengths[:, None]
    ref64 = in_float64(fo.decomp_ind1_scores, p, x, lengths, P=P)
    v2 = valid[:, :Lmax]
    assert_float_path(gf[:, :Lmax][v2], ref[v2], ref64[v2], err_msg='FULL mode')


def test_decomposed_independent1_scoring_soak():
    """Random shapes of the independent=1 scoring (MFMA path and generic fallback) against the oracle
    (FARNN_D1_SOAK_ITERS configurations, default 6; 300 ran clean)."""
    from re2nn_seq_amd import _lib, synth
    iters = int(os.environ.get('FARNN_D1_SOAK_ITERS', '6'))
    rng = np.random.RandomState(int(os.environ.get('FARNN_D1_SOAK_SEED', '31')))
    for it in range(iters):
        S, RO, K = int(rng.randint(2, 125)), int(rng.randint(1, 90)), int(rng.randint(2, 140))
        V, R, B, L = int(rng.randint(3, 80)), int(rng.randint(1, 20)), int(rng.randint(1, 40)), int(rng.randint(1, 30))
        prio = bool(rng.rand() < 0.4)
        p = {'Vgen': (rng.randn(V, R) * 0.4).astype(np.float32), 'S1': (rng.randn(S, R) * 0.3).astype(np.float32),
             'S2': (rng.randn(S, R) * 0.3).astype(np.float32), 'W': (rng.rand(S, S) < 0.1).astype(np.float32) * 0.5,
             'Cout': (rng.randn(K, RO) * 0.5).astype(np.float32), 'S1o': (rng.randn(S, RO) * 0.1).astype(np.float32),
             'S2o': (rng.randn(S, RO) * 0.1).astype(np.float32), 'Wo': None,
             'h0': np.eye(S, dtype=np.float32)[0], 'hT': (rng.rand(S) < 0.3).astype(np.float32),
             'farnn': 0, 'nl': fo.NL_CODES['tanh'], 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
        P = (np.eye(K) + (rng.rand(K, K) < 0.05) * 0.5).astype(np.float32) if prio else None
        x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
        lengths[rng.randint(0, B)] = L
        if B > 2:
            lengths[rng.randint(0, B)] = 0
        full = bool(rng.rand() < 0.3)
        h = _lib.create_decomp_ind1(p['Vgen'], p['S1'], p['S2'], p['W'], p['Cout'], p['S1o'], p['S2o'],
                                    p['h0'], p['hT'], P=P, nl='tanh', threshold=0.5, o_idx=1)
        xd, ld = _t(x).cuda(), _t(lengths).cuda()
        tags = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
        scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_FULL if full else _lib.MODE_LOCAL, tags.data_ptr(), None,
              scores.data_ptr())
        torch.cuda.synchronize()
        tag = 'iteration {}: S={} RO={} K={} V={} R={} B={} L={} prio={} full={}'.format(it, S, RO, K, V, R, B, L, prio, full)
        Lmax = max(int(lengths.max()), 1)
        ref = fo.decomp_ind1_scores(p, x, lengths, P=P)
        got = scores.cpu().numpy()
        mask = np.arange(L)[None, :] < lengths[:, None]
        m2 = mask[:, :Lmax]
        assert_float_path(got[:, :Lmax][m2], ref[m2], in_float64(fo.decomp_ind1_scores, p, x, lengths, P=P)[m2], err_msg=tag)
        assert np.isfinite(got).all(), tag
        if not full:
            assert (got[~mask] == 0).all() and (tags.cpu().numpy()[~mask] == -1).all(), tag
        want = fo.forward_local_tags(got[:, :Lmax], lengths, 0.5, 1)
        assert np.array_equal(tags.cpu().numpy()[mask], want), tag
        h.close()


# ---------------------------------------------------------------- decomposed independent=0 (a15)
def _fst_configs():
    with open(os.path.join(GOLDEN, 'decomp_fst_small.json')) as f:
        return json.load(f)


def _build_fst(g, k, cfg, meta):
    from re2nn_seq_amd.farnn.model_decompose import FARNN_S_D_W
    a = ns(independent=0, threshold=meta['threshold'], **cfg)
    torch.manual_seed(0)
    m = FARNN_S_D_W(V=g['V_in'], C=g['C_in'], S1=g['S1_in'], S2=g['S2_in'], C_wildcard=g['Cw_in'],
                    S1_wildcard=g['S1w_in'], S2_wildcard=g['S2w_in'], wildcard_wildcard=g['WW_in'],
                    final_vector=g['final_in'], start_vector=g['start_in'],
                    pretrained_word_embed=g['E_in'], priority_mat=g['priority_in'], args=a,
                    o_idx=meta['o_idx'])
    pre = 'c{}.'.format(k)
    sd = {}
    for key in g.files:
        if key.startswith(pre) and key not in (pre + 'scores', pre + 'flat_pred'):
            name = key[len(pre):]
            name = {'embedding': 'embedding.weight', 'priority_mat': 'priority_layer.priority_mat',
                    'crf_transitions': 'crf.transitions'}.get(name, name)
            sd[name] = g[key]
    names = ('S1', 'S2', 'C_embed', 'C_wildcard', 'S1_wildcard', 'S2_wildcard', 'wildcard_wildcard', 'h0', 'hT')
    shapes = {n: tuple(getattr(m, n).shape) for n in names}
    m.load_state_dict(sd)
    for n, shp in shapes.items():      # the mirror's constructor built the reference's shapes
        assert tuple(getattr(m, n).shape) == shp, n
    return m


@pytest.mark.parametrize('k', range(len(_fst_configs()['configs'])))
def test_decomposed_independent0_vs_reference(k):
    """FARNN_S_D_W (--independent 0): scores within 1e-4 of the captured reference, tags equal."""
    from re2nn_seq_amd import _lib
    meta = _fst_configs()
    cfg = meta['configs'][k]
    g = load_golden('decomp_fst_small')
    x, lengths = g['x'], g['lengths']
    m = _build_fst(g, k, cfg, meta)
    pre = 'c{}.'.format(k)
    ref_scores = g[pre + 'scores']
    Lmax = int(lengths.max())
    r = m.run(_t(x[:, :Lmax]), _t(lengths), _lib.MODE_FULL, want_scores=True)
    np.testing.assert_allclose(r['scores'].cpu().numpy(), ref_scores, rtol=1e-4, atol=1e-4)
    _, pred, true = m.forward_local(_t(x), torch.zeros_like(_t(x)), _t(lengths), train=False)
    assert np.array_equal(pred.numpy(), g[pre + 'flat_pred'])
    assert true.shape == pred.shape


def test_decomposed_independent0_ragged_vs_oracle():
    """A larger ragged batch (B=48, L=24, S=23, R=60, RW=20), LOCAL mode (fast chain path)."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(12)
    V, S, R, RW, K, B, L = 90, 23, 60, 20, 9, 48, 24
    f = lambda *shape, sc=0.3: (rng.randn(*shape) * sc).astype(np.float32)      # noqa: E731
    p = {'Vgen': f(V, R), 'C': f(K, R), 'S1': f(S, R), 'S2': f(S, R), 'Cw': f(K, RW, sc=0.2),
         'S1w': f(S, RW, sc=0.2), 'S2w': f(S, RW, sc=0.2), 'WW': (rng.rand(S, S) < 0.1).astype(np.float32) * 0.3,
         'h0': np.eye(S, dtype=np.float32)[0], 'hT': (rng.rand(S) < 0.3).astype(np.float32),
         'farnn': 0, 'nl': fo.NL_CODES['tanh'], 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    h = _lib.create_decomp_fst(p['Vgen'], p['C'], p['S1'], p['S2'], p['Cw'], p['S1w'], p['S2w'], p['WW'],
                               p['h0'], p['hT'], nl='tanh', threshold=0.5, o_idx=2)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    tags = torch.empty((B, L), dtype=torch.int32, device='cuda')
    flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), flat.data_ptr(),
          scores.data_ptr())
    torch.cuda.synchronize()
    Lmax = int(lengths.max())
    ref = fo.decomp_fst_scores(p, x, lengths)
    got = scores.cpu().numpy()[:, :Lmax]
    mask = np.arange(Lmax)[None, :] < lengths[:, None]
    np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4)
    assert np.array_equal(flat.cpu().numpy(), fo.forward_local_tags(ref, lengths, 0.5, 2))
    assert (tags.cpu().numpy()[:, :Lmax][~mask] == -1).all()


@pytest.mark.parametrize('S,R,farnn,nl,B,L', [
    (104, 50, 0, 'tanh', 12, 20),         # BASELINE configs[2] geometry: every packed row LDS-resident
    (104, 250, 2, 'relutanh', 12, 20),    # the shipped example's geometry: gated, rows streamed from L2
    (104, 100, 1, 'tanh', 9, 17),
    (200, 60, 0, 'tanh', 5, 9),           # more than 128 rows: two passes of the row phases
    (37, 300, 2, 'relu', 7, 11),          # rank >> states
    (5, 3, 1, 'none', 3, 4),
])
def test_decomposed_rows_kernel_geometries_vs_oracle(S, R, farnn, nl, B, L):
    """The rows kernel at production geometries (residency, streaming, multi-pass rows, gates) against the
    oracle: scores within 1e-4, tags equal where the margin allows."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(S * 7 + R)
    V, K = 120, 11
    p = synth.random_decomposed_params(V, S, K, R, 16, rng)
    q = {'Vgen': p['V_embed'].astype(np.float32), 'S1': p['S1'].astype(np.float32), 'S2': p['S2'].astype(np.float32),
         'W': p['wildcard_mat'].astype(np.float32), 'Cout': p['C_output_mat'].astype(np.float32),
         'h0': p['start_vector'].astype(np.float32), 'hT': p['final_vector'].astype(np.float32),
         'farnn': farnn, 'nl': fo.NL_CODES[nl], 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
    gates = None
    if farnn:
        gates = {'Wss1': rng.randn(S, S) * 0.1, 'Wrs1': rng.randn(R, S) * 0.1, 'bs1': np.full(S, 0.3)}
        if farnn == 2:
            gates.update(Wss2=rng.randn(S, S) * 0.1, Wrs2=rng.randn(R, S) * 0.1, bs2=np.full(S, 0.2))
        gates = {k: v.astype(np.float32) for k, v in gates.items()}
        q.update(gates)
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn,
                                gates=gates, sigmoid_exponent=5, nl=nl, o_idx=2)
    assert h.kernel_name(_lib.KERN_CHAIN) in ('decomp_rows_kernel', 'decomp_regs_kernel')
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, flat.data_ptr(), scores.data_ptr())
    torch.cuda.synchronize()
    Lmax = int(lengths.max())
    ref = fo.decomp_ifst_scores(q, x, lengths)
    mask = np.arange(Lmax)[None, :] < lengths[:, None]
    got = scores.cpu().numpy()[:, :Lmax]
    np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4)
    # tags: equal wherever the oracle's decision margin exceeds the tolerance
    rt = fo.forward_local_tags(ref, lengths, 0.5, 2)
    refc = ref.copy(); refc[..., -1] = np.minimum(refc[..., -1], 0.5)
    top2 = np.sort(refc[:, :Lmax][mask], axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-3
    assert np.array_equal(flat.cpu().numpy()[safe], rt[safe])


def test_materialised_modes_run_on_dense_blocks():
    """max-times semiring / independent=1 with farnn=0: the per-step matrix depends on the word only, so the
    handle serves these modes from a dense per-word block table through the chain kernel."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(3)
    V, S, K, R, B, L = 60, 21, 7, 30, 9, 12
    p = synth.random_decomposed_params(V, S, K, R, 16, rng)
    q = {'Vgen': p['V_embed'].astype(np.float32), 'S1': p['S1'].astype(np.float32), 'S2': p['S2'].astype(np.float32),
         'W': p['wildcard_mat'].astype(np.float32), 'Cout': p['C_output_mat'].astype(np.float32),
         'h0': p['start_vector'].astype(np.float32), 'hT': p['final_vector'].astype(np.float32),
         'farnn': 0, 'nl': fo.NL_CODES['relu'], 'semiring': fo.SEMIRING_MAX, 'sig_k': 5}
    h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], nl='relu',
                                semiring='max', o_idx=2)
    assert h.kernel_name(_lib.KERN_CHAIN).startswith('chain_kernel')
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    scores = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, None, scores.data_ptr())
    torch.cuda.synchronize()
    Lmax = int(lengths.max())
    ref = fo.decomp_ifst_scores(q, x, lengths)
    mask = np.arange(Lmax)[None, :] < lengths[:, None]
    np.testing.assert_allclose(scores.cpu().numpy()[:, :Lmax][mask], ref[mask], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('farnn,semiring,R', [(0, 'sum', 50), (2, 'sum', 40), (1, 'max', 12), (0, 'sum', 90)])
def test_one_handle_serves_varying_batch_shapes(farnn, semiring, R):
    """the workspace is a capacity, strided with the call's (B, L): a loop whose batches vary in length
    (the decomposed mirrors clip every batch to lengths.max()) allocates once, and results do not depend on what an
    earlier, differently shaped call left behind (pad columns, longer rows).  Register kernel (farnn 0, rank <= 64), rows
    kernel (gated / rank 90) and the generic kernel (gated max semiring) alike; S = 21 has pad columns (SP = 24)."""
    from re2nn_seq_amd import _lib, synth
    rng = np.random.RandomState(11 + farnn)
    V, S, K = 80, 21, 7
    p = synth.random_decomposed_params(V, S, K, R, 16, rng, contractive=True)
    f = lambda a: np.asarray(a, np.float32)                    # noqa: E731
    q = {'Vgen': f(p['V_embed']), 'S1': f(p['S1']), 'S2': f(p['S2']), 'W': f(p['wildcard_mat']), 'Cout': f(p['C_output_mat']),
         'h0': f(p['start_vector']), 'hT': f(p['final_vector']), 'farnn': farnn, 'nl': fo.NL_CODES['tanh'],
         'semiring': fo.SEMIRING_MAX if semiring == 'max' else fo.SEMIRING_SUM, 'sig_k': 5}
    gates = None
    if farnn:
        gates = {'Wss1': f(rng.randn(S, S) * 0.1), 'Wrs1': f(rng.randn(R, S) * 0.1), 'bs1': f(np.full(S, 0.3))}
        if farnn == 2:
            gates.update(Wss2=f(rng.randn(S, S) * 0.1), Wrs2=f(rng.randn(R, S) * 0.1), bs2=f(np.full(S, 0.2)))
        q.update(gates)
    h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], farnn=farnn, gates=gates,
                                sigmoid_exponent=5, nl='tanh', semiring=semiring, o_idx=2)
    for B, L in [(9, 20), (5, 7), (12, 20), (3, 33), (9, 20)]:
        x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
        xd, ld = _t(x).cuda(), _t(lengths).cuda()
        scores = torch.full((B, L, K), 7.0, dtype=torch.float32, device='cuda')
        flat = torch.empty((int(lengths.sum()),), dtype=torch.int64, device='cuda')
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, flat.data_ptr(), scores.data_ptr())
        torch.cuda.synchronize()
        ref = fo.decomp_ifst_scores(q, x, lengths)
        Lmax = ref.shape[1]
        mask = np.arange(Lmax)[None, :] < lengths[:, None]
        got = scores.cpu().numpy()[:, :Lmax]
        np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4, err_msg='B={} L={}'.format(B, L))
        assert (scores.cpu().numpy()[:, :Lmax][~mask] == 0).all()
    h.close()


@pytest.mark.parametrize('add_nl', ['none', 'relu', 'tanh', 'sigmoid', 'relutanh'])
@pytest.mark.parametrize('normalize', ['none', 'l2-rank', 'l1-rank', 'l1', 'l2'])
def test_word_table_and_normalisation_folded_on_the_device(add_nl, normalize):
    """farnn_decomp_ifst_create_folded computes Vgen = V_embed*beta + nl_add(E@G)*(1-beta)
    (model_decompose.py:222-241) and the --normalize_automata scaling (init_params.py:285-297; all four modes of utils.get_average:
    per-rank column norms, numpy's matrix 1-norm, the spectral norm) on the device.
    Against the host statement of both (the oracle's generalized_vocab_table on factors normalised with
    utils.get_average) through the regular creator: same scores."""
    from re2nn_seq_amd import _lib, synth
    from re2nn_seq_amd.utils import get_average
    rng = np.random.RandomState(5)
    V, S, K, R, D, B, L = 90, 21, 7, 30, 12, 8, 14
    p = synth.random_decomposed_params(V, S, K, R, D, rng, contractive=True)
    Vemb, S1, S2, E = (np.asarray(p[k], np.float64) for k in ('V_embed', 'S1', 'S2', 'embed'))
    G = np.linalg.pinv(E) @ Vemb                        # the bridge of the UN-normalised V_embed (:73-76)
    beta = np.full(R, 0.6)
    Vn, S1n, S2n, Gn = Vemb, S1, S2, G
    if normalize != 'none':                             # init_params.py:285-297 on the host, float64 like the reference
        va, s1a, s2a = get_average(Vemb, normalize), get_average(S1, normalize), get_average(S2, normalize)
        fac = np.float_power(va * s1a * s2a, 1 / 3)
        Vn, S1n, S2n = Vemb * (fac / va), S1 * (fac / s1a), S2 * (fac / s2a)
        Gn = np.linalg.pinv(E) @ Vn                     # what the mirror's constructor computes from the normalised table
    vg = fo.generalized_vocab_table(Vn, E, Gn, beta, add_nl=fo.NL_CODES[add_nl])
    f = lambda a: np.asarray(a, np.float32)             # noqa: E731
    common = dict(nl='tanh', o_idx=2)
    W, Cout, h0, hT = f(p['wildcard_mat']), f(p['C_output_mat']), f(p['start_vector']), f(p['final_vector'])
    h_host = _lib.create_decomp_ifst(vg, f(S1n), f(S2n), W, Cout, h0, hT, **common)
    h_dev = _lib.create_decomp_ifst_folded(f(Vemb), f(E), f(G), f(beta), f(S1), f(S2), W, Cout, h0, hT, add_nl=add_nl,
                                           normalize=normalize, **common)
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    xd, ld = _t(x).cuda(), _t(lengths).cuda()
    out = []
    for h in (h_host, h_dev):
        sc = torch.empty((B, L, K), dtype=torch.float32, device='cuda')
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, None, None, sc.data_ptr())
        torch.cuda.synchronize()
        out.append(sc.cpu().numpy())
        h.close()
    assert np.abs(out[0]).max() > 0.1
    np.testing.assert_allclose(out[1], out[0], rtol=2e-5, atol=2e-6)
    with pytest.raises(_lib.FarnnError):                 # a mode the header does not name
        _lib.NORM['bogus'] = 9
        try:
            _lib.create_decomp_ifst_folded(f(Vemb), f(E), f(G), f(beta), f(S1), f(S2), W, Cout, h0, hT, normalize='bogus')
        finally:
            del _lib.NORM['bogus']


def test_decomposed_random_geometries_vs_oracle():
    """Short form of tests/soak_decomp_shapes.py (150 random geometries ran clean): states x rank x gates x CRF x
    batch x length drawn at random -- register kernel, the rows kernel's register forms, the LDS + L2 path, score tiles and
    Viterbi layouts at shapes no fixed test names; scores within 1e-4 of the oracle, Viterbi bit-exact on the GPU's scores."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('soak_decomp_shapes', os.path.join(os.path.dirname(__file__), 'soak_decomp_shapes.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(n=int(os.environ.get('FARNN_SOAK_SHAPES', '30')), seed=7, verbose=True) == 0

"""Pin the CPU oracle (oracle/farnn_oracle.py) against outputs captured from the imported
reference classes (tests/golden/make_golden.py).  The reference has no golden vectors of its
own, so these fixtures are the pin.

Tolerances: onehot 0/1 automata -> bit-exact scores and tags; tanh / dense-valued and
decomposed paths -> scores within 1e-4 (north_star), tags equal.
"""
import json
import os

import numpy as np
import pytest

from oracle import farnn_oracle as fo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def _check(sc, ref_scores, exact):
    if exact:
        assert np.array_equal(sc, ref_scores)
    else:
        np.testing.assert_allclose(sc, ref_scores, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('nl', ['none', 'relu', 'tanh', 'relutanh'])
@pytest.mark.parametrize('mode', ['sum', 'max'])
@pytest.mark.parametrize('up', [0, 1])
def test_ifst_small(nl, mode, up):
    g = _load('ifst_small')
    x, l, o_idx = g['x'], g['lengths'], int(g['o_idx'])
    C = g['O'].shape[0]
    P = fo.expand_priority(C, g['priority']) if up else None
    sc = fo.onehot_ifst_scores(g['T'], g['W'], g['O'], g['h0'], g['hT'], x, l,
                               fo.NL_CODES[nl], fo.SEMIRING_MAX if mode == 'max' else fo.SEMIRING_SUM, P)
    key = '{}.{}.p{}.'.format(nl, mode, up)
    _check(sc, g[key + 'scores'], exact=nl in ('none', 'relu'))
    assert np.array_equal(fo.forward_local_tags(sc, l, 0.5, o_idx), g[key + 'flat_pred'])
    assert np.array_equal(fo.decode_argmax(sc, 0.5, o_idx), g[key + 're_pred'])   # pads included
    clamped = sc.copy()            # forward_RE returns the clamped clone (model_onehot.py:153-154)
    clamped[..., -1] = np.minimum(clamped[..., -1], np.float32(0.5))
    _check(clamped, g[key + 're_scores'], exact=nl in ('none', 'relu'))


@pytest.mark.parametrize('nl', ['none', 'tanh'])
def test_ifst_dense_values(nl):
    g = _load('ifst_dense')
    x, l, o_idx = g['x'], g['lengths'], int(g['o_idx'])
    sc = fo.onehot_ifst_scores(g['T'], g['W'], g['O'], g['h0'], g['hT'], x, l, fo.NL_CODES[nl])
    _check(sc, g[nl + '.scores'], exact=False)
    thr = float(g['threshold'])
    assert np.array_equal(fo.forward_local_tags(sc, l, thr, o_idx), g[nl + '.flat_pred'])
    assert np.array_equal(fo.decode_argmax(sc, thr, o_idx), g[nl + '.re_pred'])


@pytest.mark.parametrize('mode', ['sum', 'max'])
@pytest.mark.parametrize('up', [0, 1])
def test_fst4_small(mode, up):
    g = _load('fst4_small')
    x, l, o_idx = g['x'], g['lengths'], int(g['o_idx'])
    C = g['W4'].shape[0]
    P = fo.expand_priority(C, g['priority']) if up else None
    sc = fo.onehot_fst4_scores(g['T4'], g['W4'], g['h0'], g['hT'], x, l,
                               fo.SEMIRING_MAX if mode == 'max' else fo.SEMIRING_SUM, P)
    key = '{}.p{}.'.format(mode, up)
    assert np.array_equal(sc, g[key + 'scores'])
    assert np.array_equal(fo.forward_local_tags(sc, l, 0.5, o_idx), g[key + 'flat_pred'])
    assert np.array_equal(fo.decode_argmax(sc, 0.5, o_idx), g[key + 're_pred'])


@pytest.mark.parametrize('mode', ['sum', 'max'])
@pytest.mark.parametrize('ind', [1, 2])
def test_ind1_small(mode, ind):
    g = _load('ind1_small')
    x, l, o_idx = g['x'], g['lengths'], int(g['o_idx'])
    sc = fo.onehot_ind1_scores(g['T'], g['W'], g['Oten'], g['h0'], g['hT'], x, l,
                               fo.SEMIRING_MAX if mode == 'max' else fo.SEMIRING_SUM,
                               mask_by_output=(ind == 2))
    key = '{}.ind{}.'.format(mode, ind)
    assert np.array_equal(sc, g[key + 'scores'])
    assert np.array_equal(fo.forward_local_tags(sc, l, 0.5, o_idx), g[key + 'flat_pred'])


@pytest.mark.parametrize('which', ['default', 'random'])
def test_crf_viterbi(which):
    g = _load('crf_small')
    feats, l = g['feats'], g['lengths']
    paths = fo.viterbi_paths(feats, l, g[which + '.tr'])
    ref = g[which + '.paths']
    for b in range(feats.shape[0]):
        n = int(l[b])
        assert np.array_equal(paths[b, :n], ref[b, :n]), b
    if which == 'default':
        assert np.array_equal(g['default.tr'], fo.crf_default_transitions(feats.shape[2] - 2))


def decomp_params_from_fixture(g, k, cfg):
    """Build the oracle's parameter dict for decomposed config k of decomp_small."""
    pre = 'c{}.'.format(k)
    p = {
        'S1': g[pre + 'S1'], 'S2': g[pre + 'S2'], 'W': g[pre + 'wildcard_mat'],
        'Cout': g[pre + 'C_output_mat'], 'h0': g[pre + 'h0'], 'hT': g[pre + 'hT'],
        'farnn': cfg.get('farnn', 0), 'nl': fo.NL_CODES[cfg.get('update_nonlinear', 'none')],
        'semiring': fo.SEMIRING_MAX if cfg.get('train_mode', 'sum') == 'max' else fo.SEMIRING_SUM,
        'sig_k': cfg.get('sigmoid_exponent', 5),
    }
    p['Vgen'] = fo.generalized_vocab_table(
        g[pre + 'V_embed'], g[pre + 'embedding'], g[pre + 'embed_r_generalized'],
        g[pre + 'beta_vec'], fo.NL_CODES[cfg.get('additional_nonlinear', 'none')])
    for kk in ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2'):
        if pre + kk in g.files:
            p[kk] = g[pre + kk]
    return p


def _decomp_configs():
    with open(os.path.join(GOLDEN, 'decomp_small.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('k', range(len(_decomp_configs()['configs'])))
def test_decomposed_small(k):
    meta = _decomp_configs()
    cfg = meta['configs'][k]
    g = _load('decomp_small')
    x, l = g['x'], g['lengths']
    pre = 'c{}.'.format(k)
    p = decomp_params_from_fixture(g, k, cfg)
    P = g[pre + 'priority_mat'] if cfg.get('use_priority', 0) else None
    sc = fo.decomp_ifst_scores(p, x, l, P)
    ref = g[pre + 'scores']
    assert sc.shape == ref.shape
    np.testing.assert_allclose(sc, ref, rtol=1e-4, atol=1e-4)
    tr = g[pre + 'crf_transitions'] if cfg.get('use_crf', 0) else None
    tags = fo.forward_local_tags(sc, l, meta['threshold'], meta['o_idx'], tr)
    assert np.array_equal(tags, g[pre + 'flat_pred'])


def test_atis_scale_ifst():
    """ATIS-scale i-FST fixture: inputs are regenerated from the stored seed (the tensors are
    too big to commit); tags and sampled score rows come from the reference."""
    from re2nn_seq_amd import synth
    g = _load('atis_ifst')
    V, S, C, B, L = [int(v) for v in g['dims']]
    rng = np.random.RandomState(int(g['seed']))
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
    x = g['x'].astype(np.int64); l = g['lengths'].astype(np.int64)
    sc = fo.onehot_ifst_scores(T, W, O, h0, hT, x, l)
    rows = g['sample_rows']
    assert np.array_equal(sc[rows], g['sample_scores'])
    assert np.array_equal(fo.decode_argmax(sc, 0.5, 0), g['tags'].astype(np.int64))
    assert np.array_equal(fo.forward_local_tags(sc, l, 0.5, 0), g['flat_pred'].astype(np.int64))
    assert int((g['flat_pred'] != 0).sum()) > 100          # rules do fire in this fixture


def decomp_ind1_params_from_fixture(g, k, cfg):
    pre = 'c{}.'.format(k)
    p = {
        'S1': g[pre + 'S1'], 'S2': g[pre + 'S2'], 'W': g[pre + 'wildcard_mat'],
        'Cout': g[pre + 'C_output'], 'S1o': g[pre + 'S1_output'], 'S2o': g[pre + 'S2_output'], 'Wo': None,
        'h0': g[pre + 'h0'], 'hT': g[pre + 'hT'],
        'farnn': cfg.get('farnn', 0), 'nl': fo.NL_CODES[cfg.get('update_nonlinear', 'none')],
        'semiring': fo.SEMIRING_MAX if cfg.get('train_mode', 'sum') == 'max' else fo.SEMIRING_SUM,
        'sig_k': cfg.get('sigmoid_exponent', 5),
    }
    p['Vgen'] = fo.generalized_vocab_table(
        g[pre + 'V_embed'], g[pre + 'embedding'], g[pre + 'embed_r_generalized'],
        g[pre + 'beta_vec'], fo.NL_CODES[cfg.get('additional_nonlinear', 'none')])
    for kk in ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2'):
        if pre + kk in g.files:
            p[kk] = g[pre + kk]
    return p


def _ind1_configs():
    with open(os.path.join(GOLDEN, 'decomp_ind1_small.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('k', range(len(_ind1_configs()['configs'])))
def test_decomposed_independent1_small(k):
    """FARNN_S_D_W_I (--method decompose --independent 1), SURVEY.md 8a row a15."""
    meta = _ind1_configs()
    cfg = meta['configs'][k]
    g = _load('decomp_ind1_small')
    x, l = g['x'], g['lengths']
    pre = 'c{}.'.format(k)
    p = decomp_ind1_params_from_fixture(g, k, cfg)
    P = g[pre + 'priority_mat'] if cfg.get('use_priority', 0) else None
    sc = fo.decomp_ind1_scores(p, x, l, P)
    ref = g[pre + 'scores']
    assert sc.shape == ref.shape
    np.testing.assert_allclose(sc, ref, rtol=1e-4, atol=1e-4)
    tr = g[pre + 'crf_transitions'] if cfg.get('use_crf', 0) else None
    tags = fo.forward_local_tags(sc, l, meta['threshold'], meta['o_idx'], tr)
    assert np.array_equal(tags, g[pre + 'flat_pred'])


def decomp_fst_params_from_fixture(g, k, cfg):
    pre = 'c{}.'.format(k)
    p = {
        'S1': g[pre + 'S1'], 'S2': g[pre + 'S2'], 'C': g[pre + 'C_embed'], 'Cw': g[pre + 'C_wildcard'],
        'S1w': g[pre + 'S1_wildcard'], 'S2w': g[pre + 'S2_wildcard'], 'WW': g[pre + 'wildcard_wildcard'],
        'h0': g[pre + 'h0'], 'hT': g[pre + 'hT'],
        'farnn': cfg.get('farnn', 0), 'nl': fo.NL_CODES[cfg.get('update_nonlinear', 'none')],
        'semiring': fo.SEMIRING_MAX if cfg.get('train_mode', 'sum') == 'max' else fo.SEMIRING_SUM,
        'sig_k': cfg.get('sigmoid_exponent', 5),
    }
    p['Vgen'] = fo.generalized_vocab_table(
        g[pre + 'V_embed'], g[pre + 'embedding'], g[pre + 'embed_r_generalized'],
        g[pre + 'beta_vec'], fo.NL_CODES[cfg.get('additional_nonlinear', 'none')])
    for kk in ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2'):
        if pre + kk in g.files:
            p[kk] = g[pre + kk]
    return p


def _fst_configs():
    with open(os.path.join(GOLDEN, 'decomp_fst_small.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('k', range(len(_fst_configs()['configs'])))
def test_decomposed_independent0_small(k):
    """FARNN_S_D_W (--method decompose --independent 0), SURVEY.md 8a row a15."""
    meta = _fst_configs()
    cfg = meta['configs'][k]
    g = _load('decomp_fst_small')
    x, l = g['x'], g['lengths']
    pre = 'c{}.'.format(k)
    p = decomp_fst_params_from_fixture(g, k, cfg)
    P = g[pre + 'priority_mat'] if cfg.get('use_priority', 0) else None
    sc = fo.decomp_fst_scores(p, x, l, P)
    ref = g[pre + 'scores']
    assert sc.shape == ref.shape
    np.testing.assert_allclose(sc, ref, rtol=1e-4, atol=1e-4)
    tr = g[pre + 'crf_transitions'] if cfg.get('use_crf', 0) else None
    tags = fo.forward_local_tags(sc, l, meta['threshold'], meta['o_idx'], tr)
    assert np.array_equal(tags, g[pre + 'flat_pred'])

"""Metrics against the reference's own outputs; CLI flag parity, sanity asserts and the
`--args_path` reload of one of the reference's shipped example `.res` files."""
import json
import os

import numpy as np
import pytest

from re2nn_seq_amd import main as cli
from re2nn_seq_amd.metrics.metrics import eval_seq_token, get_ner_BIO, get_ner_fmeasure
from util import GOLDEN


def test_metrics_match_reference():
    with open(os.path.join(GOLDEN, 'metrics_small.json')) as f:
        m = json.load(f)
    i2s = {int(k): v for k, v in m['i2s'].items()}
    tok = eval_seq_token(np.array(m['pred']), np.array(m['true']), o_idx=m['o_idx'])
    np.testing.assert_allclose(tok, m['token'], rtol=0, atol=1e-12)
    acc, p, r, f, cls = get_ner_fmeasure(m['true'], m['pred'], label_type='BIO', i2s=i2s, all_class=True)
    np.testing.assert_allclose([acc, p, r, f], m['entity'], rtol=0, atol=1e-12)
    assert set(cls) == set(m['per_class'])
    for k, v in m['per_class'].items():
        np.testing.assert_allclose(cls[k], v, rtol=0, atol=1e-12)


def test_bio_span_quirks():
    spans = get_ner_BIO(['o', 'b-x', 'i-x', 'i-y', 'b-y', 'o', 'b-z'])
    assert spans == ['[1,2]X', '[4,4]Y', '[6]Z']      # I- of another type closes; last span open-ended


def test_cli_flags_match_reference_parser():
    with open(os.path.join(GOLDEN, 'cli_flags.json')) as f:
        ref = json.load(f)['flags']
    parser = cli.build_parser()
    ours = {a.dest: a for a in parser._actions if a.dest != 'help'}
    tmap = {'str': str, 'int': int, 'float': float}
    for name, typ, default in ref:
        assert name in ours, name
        assert ours[name].type is tmap[typ], name
        assert ours[name].default == default, name
    assert set(ours) - {n for n, _, _ in ref} == {'data_dir', 'model_dir'}


def _args(*extra):
    base = ['--method', 'onehot', '--rand_constant', '0', '--normalize_automata', 'none',
            '--dataset', 'ATIS-BIO', '--epoch', '0', '--train_portion', '0']
    return cli.parse_args(base + list(extra))


def test_cli_sanity_asserts():
    a, p = _args()
    cli.check_args(a, p)
    for bad in (['--rand_constant', '1e-5'],                 # onehot => rand_constant == 0 (:175-176)
                ['--normalize_automata', 'l2'],              # normalisation => decompose (:150-151)
                ['--epoch', '3'],                            # train_portion 0 => epoch 0 (:147-148)
                ['--rank', '50'],                            # rank enumeration (:136)
                ['--dataset', 'MITR-BIO'],                   # dataset enumeration (:179)
                ['--update_nonlinear', 'gelu']):
        a, p = _args(*bad)
        with pytest.raises(AssertionError):
            cli.check_args(a, p)
    a, p = _args('--method', 'baseline')
    cli.check_args(a, p)
    with pytest.raises(NotImplementedError):
        cli.dispatch(a)


def test_args_path_reload_of_reference_example():
    path = os.path.join(GOLDEN, 'example_ATIS-ZH_0pct.res')
    res = cli.load_res(path)
    assert res['args'].rank == 250 and res['args'].farnn == 2 and res['args'].use_crf == 1
    assert abs(res['res'].best_dev_test_results['entity-level'][3] - 0.7503) < 1e-4
    assert len(res['logger'].record) > 0
    a, _ = _args()
    merged = cli.merge_saved_args(a, path)
    assert merged.run == 'final_222' and merged.method == 'decompose' and merged.independent == 2
    assert merged.data_dir == '../data/'


def test_vectorised_entity_metrics_equal_the_string_automaton():
    """get_ner_fmeasure's array form (bio_spans) against the token-by-token string form that follows the
    reference (metrics.py:184-229), incl. I- tags of another type, spans left open at the end of the list,
    two ids with one spelling, and the per-class dict's insertion order."""
    from re2nn_seq_amd.metrics import metrics as M
    rng = np.random.RandomState(0)
    i2s = {0: 'o'}
    for t in ['city', 'time', 'airline', 'x']:
        i2s[len(i2s)] = 'B-' + t
        i2s[len(i2s)] = 'I-' + t
    i2s[len(i2s)] = 'o'
    nl = len(i2s)
    for trial in range(300):
        n = rng.randint(1, 60)
        gold = rng.randint(0, nl, n)
        pred = np.where(rng.rand(n) < rng.rand(), gold, rng.randint(0, nl, n))
        a = M.get_ner_fmeasure(gold, pred, i2s=i2s, all_class=True)
        b = M._get_ner_fmeasure_strings(gold, pred, i2s, True)
        assert a[:4] == b[:4] and a[4] == b[4] and list(a[4]) == list(b[4]), (gold, pred)
    # a B- label with an empty type takes the string path
    i2s[nl] = 'B-'
    gold = rng.randint(0, nl + 1, 50)
    assert M.get_ner_fmeasure(gold, gold, i2s=i2s, all_class=True) == M._get_ner_fmeasure_strings(gold, gold, i2s, True)

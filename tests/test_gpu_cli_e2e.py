"""End-to-end through the reference's command line on a synthetic, schema-identical data tree:
dataset.pkl + automaton pickle -> loaders -> model mirror -> C-ABI -> HIP kernels -> metrics ->
.res file.  Expected predictions come from the CPU oracle run on the same padded batches."""
import os

import numpy as np
import pytest
import torch

from oracle import farnn_oracle as fo
from re2nn_seq_amd import main as cli
from re2nn_seq_amd import synth
from re2nn_seq_amd.metrics.metrics import eval_seq_token, get_ner_fmeasure
from re2nn_seq_amd.utils import pad_dataset_1
from re2nn_seq_amd.wfa import fsa_to_tensor as f2t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('data'))
    return synth.write_dataset_tree(root, dataset='ATIS-BIO', seed=4)


def _expected(tree, split, L, score_fn):
    dset = tree['dset']
    t2i = dict(dset['t2i']); t2i['<pad>'] = len(t2i)
    s2i, i2s = dset['s2i'], dset['i2s']
    q, _, lens = pad_dataset_1(dset['query_' + split], L, t2i['<pad>'])
    s, _, _ = pad_dataset_1(dset['intent_' + split], L, s2i['o'])
    x, lengths, gold = np.stack(q), np.array(lens), np.stack(s)
    pred = fo.forward_local_tags(score_fn(t2i, s2i, x, lengths), lengths, 0.5, s2i['o'])
    true = fo.flatten(gold, lengths)
    tok = eval_seq_token(pred, true, o_idx=s2i['o'])
    ent = get_ner_fmeasure(true, pred, i2s=i2s, all_class=True)
    return list(tok), list(ent)


@pytest.mark.parametrize('independent', [2, 1, 0])
def test_onehot_cli_matches_oracle(tree, tmp_path, independent):
    L = 12
    argv = ['--dataset', 'ATIS-BIO', '--method', 'onehot', '--independent', str(independent),
            '--automata_path', tree['paths']['ID{}'.format(independent)],
            '--normalize_automata', 'none', '--rand_constant', '0', '--bz', '10', '--seq_max_len', str(L),
            '--epoch', '0', '--train_portion', '0', '--data_dir', tree['paths']['data_dir'],
            '--model_dir', str(tmp_path), '--run', 'e2e']
    results, stats, res_path = cli.main(argv)
    automaton = tree['automaton']

    def score_fn(t2i, s2i, x, lengths):
        if independent == 2:
            T, _, W, O, _, fin, sta, _ = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i)
            return fo.onehot_ifst_scores(T, W, O, sta, fin, x, lengths)
        if independent == 1:
            T, _, W, Oten, _, fin, sta, _ = f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i)
            return fo.onehot_ind1_scores(T, W, Oten, sta, fin, x, lengths)
        T4, _, W4, _, fin, sta, _ = f2t.dfa_to_tensor_slot_new_wildcard(automaton, t2i, s2i)
        return fo.onehot_fst4_scores(T4, W4, sta, fin, x, lengths)

    for split in ('train', 'dev', 'test'):
        tok, ent = _expected(tree, split, L, score_fn)
        assert results[split]['token-level'] == tok
        assert results[split]['entity-level'] == ent
        assert stats[split]['tokens'] > 0
    assert results['test']['token-level'][3] > 0.3          # the planted rules do fire
    saved = cli.load_res(res_path)
    assert saved['args'].independent == independent and saved['res'].best_dev_results == results['dev']


def test_decompose_cli_matches_oracle(tree, tmp_path):
    L = 12
    argv = ['--dataset', 'ATIS-BIO', '--method', 'decompose', '--independent', '2',
            '--automata_path', tree['paths']['IIID'], '--rank', '100', '--seed', '1', '--beta', '1.0',
            '--embed_dim', '16', '--normalize_automata', 'none', '--rand_constant', '0',
            '--update_nonlinear', 'none', '--bz', '9', '--seq_max_len', str(L), '--epoch', '0',
            '--train_portion', '0', '--data_dir', tree['paths']['data_dir'], '--model_dir', str(tmp_path)]
    results, stats, _ = cli.main(argv)
    args, _ = cli.parse_args(argv)

    # parity, not a smoke test: the oracle's decomposed recurrence fed by the build's own loader (the same factors
    # the CLI hands to the model mirror; beta = 1 makes the word table V_embed itself, model_decompose.py:222-241)
    def score_fn(t2i, s2i, x, lengths):
        from re2nn_seq_amd.init_params import get_init_params_seq_independent_single
        (V, S1, S2, _, W, _, fin, sta, _, Cout, _) = get_init_params_seq_independent_single(
            args, s2i, t2i, data_dir=tree['paths']['data_dir'])
        q = {'Vgen': V.astype(np.float32), 'S1': S1.astype(np.float32), 'S2': S2.astype(np.float32),
             'W': W.astype(np.float32), 'Cout': Cout.astype(np.float32), 'h0': sta.astype(np.float32),
             'hT': fin.astype(np.float32), 'farnn': 0, 'nl': fo.NL_NONE, 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
        sc = fo.decomp_ifst_scores(q, x, lengths)
        # the decision margins of this near-integer model are far above float noise: equality is meaningful
        c = sc.copy(); c[..., -1] = np.minimum(c[..., -1], 0.5)
        m = np.arange(sc.shape[1])[None, :] < lengths[:, None]
        top2 = np.sort(c[m], axis=1)[:, -2:]
        assert (top2[:, 1] - top2[:, 0]).min() > 1e-2
        return sc

    for split in ('train', 'dev', 'test'):
        tok, ent = _expected(tree, split, L, score_fn)
        assert results[split]['token-level'] == tok           # every flat prediction equal -> identical counts
        assert results[split]['entity-level'] == ent
    assert results['test']['token-level'][3] > 0.3


def test_decompose_independent1_cli_matches_oracle(tree, tmp_path):
    """--method decompose --independent 1 (FARNN_S_D_W_I): IID pickle -> loader -> mirror -> HIP."""
    L = 12
    argv = ['--dataset', 'ATIS-BIO', '--method', 'decompose', '--independent', '1',
            '--automata_path', tree['paths']['IID'], '--rank', '100', '--rank_wildcard', '70', '--seed', '1',
            '--beta', '1.0', '--embed_dim', '16', '--normalize_automata', 'none', '--rand_constant', '0',
            '--update_nonlinear', 'none', '--bz', '9', '--seq_max_len', str(L), '--epoch', '0',
            '--train_portion', '0', '--data_dir', tree['paths']['data_dir'], '--model_dir', str(tmp_path)]
    results, stats, _ = cli.main(argv)
    args, _ = cli.parse_args(argv)

    # the oracle's independent=1 recurrence + scoring fed by the build's own loader
    def score_fn(t2i, s2i, x, lengths):
        from re2nn_seq_amd.init_params import get_init_params_seq_independent
        (V, S1, S2, _, W, Wo, fin, sta, _, Cout, S1o, S2o) = get_init_params_seq_independent(
            args, s2i, t2i, data_dir=tree['paths']['data_dir'])
        f = lambda a: np.asarray(a, np.float32)                    # noqa: E731
        q = {'Vgen': f(V), 'S1': f(S1), 'S2': f(S2), 'W': f(W), 'Cout': f(Cout), 'S1o': f(S1o), 'S2o': f(S2o),
             'h0': f(sta), 'hT': f(fin), 'farnn': 0, 'nl': fo.NL_NONE, 'semiring': fo.SEMIRING_SUM, 'sig_k': 5}
        sc = fo.decomp_ind1_scores(q, x, lengths)
        c = sc.copy(); c[..., -1] = np.minimum(c[..., -1], 0.5)
        m = np.arange(sc.shape[1])[None, :] < lengths[:, None]
        top2 = np.sort(c[m], axis=1)[:, -2:]
        assert (top2[:, 1] - top2[:, 0]).min() > 1e-4     # float32 noise here is ~1e-6: equality is meaningful
        return sc

    for split in ('train', 'dev', 'test'):
        tok, ent = _expected(tree, split, L, score_fn)
        assert results[split]['token-level'] == tok
        assert results[split]['entity-level'] == ent
    assert results['test']['token-level'][3] > 0.3
    assert stats['test']['tokens'] > 0


def test_predict_by_RE_scores_and_cache(tree, tmp_path):
    """RE.predict_by_RE: unflattened predictions + scores incl. pad positions, 0.99 -> 1.0 fix-up,
    cached as <automata_path>.re.score (reference RE.py:77-192)."""
    from re2nn_seq_amd.RE import predict_by_RE
    L = 12
    args, parser = cli.parse_args(
        ['--dataset', 'ATIS-BIO', '--method', 'onehot', '--independent', '2', '--normalize_automata', 'none',
         '--rand_constant', '0', '--bz', '7', '--seq_max_len', str(L), '--epoch', '0', '--train_portion', '0'])
    args.re_automata_path = tree['paths']['ID2']
    cache = tree['paths']['ID2'] + '.re.score'
    if os.path.exists(cache):
        os.remove(cache)
    out = predict_by_RE(args, data_dir=tree['paths']['data_dir'])
    assert os.path.exists(cache)
    dset = tree['dset']
    t2i = dict(dset['t2i']); t2i['<pad>'] = len(t2i)
    s2i = dset['s2i']
    T, _, W, O, _, fin, sta, _ = f2t.dfa_to_tensor_slot_single_wildcard(tree['automaton'], t2i, s2i)
    for k, split in enumerate(('train', 'dev', 'test')):
        q, _, lens = pad_dataset_1(dset['query_' + split], L, t2i['<pad>'])
        x, lengths = np.stack(q), np.array(lens)
        sc = fo.onehot_ifst_scores(T, W, O, sta, fin, x, lengths)
        assert np.array_equal(out[k].numpy(), fo.decode_argmax(sc, 0.99, s2i['o']))       # threshold 0.99
        ref = sc.copy()
        ref[..., -1] = np.minimum(ref[..., -1], np.float32(0.99))
        ref[ref == np.float32(0.99)] = 1.0
        assert np.array_equal(out[3 + k].numpy(), ref)
    again = predict_by_RE(args, data_dir=tree['paths']['data_dir'])                       # cache hit
    assert torch.equal(again[0], out[0])


def test_decompose_cli_trains_for_two_epochs(tree, tmp_path):
    """--epoch 2 with a training portion: the epoch loop (reference train_decompose.py:161-221) runs on the HIP
    training step, the loss goes down and the evaluations after each epoch use the updated weights."""
    L = 12
    argv = ['--dataset', 'ATIS-BIO', '--method', 'decompose', '--independent', '2',
            '--automata_path', tree['paths']['IIID'], '--rank', '100', '--seed', '1', '--beta', '0.9',
            '--embed_dim', '16', '--normalize_automata', 'none', '--rand_constant', '0',
            '--update_nonlinear', 'tanh', '--bz', '9', '--seq_max_len', str(L), '--epoch', '2', '--lr', '0.01',
            '--train_portion', '1.0', '--data_dir', tree['paths']['data_dir'], '--model_dir', str(tmp_path)]
    results, stats, res_path = cli.main(argv)
    steps = stats['train_step']
    assert len(steps) == 2 and all(s['tokens'] > 0 and s['tokens_per_s'] > 0 for s in steps)
    saved = cli.load_res(res_path)
    assert saved['args'].epoch == 2
    losses = [float(line.split('LOSS:')[1]) for line in saved['logger'].record if 'LOSS:' in line]
    assert len(losses) == 2 and losses[1] < losses[0]
    assert sum('| 2 |' in line or 'Epoch: 2' in line for line in saved['logger'].record) >= 1


def test_decompose_cli_trains_with_the_crf(tree, tmp_path):
    """--use_crf 1 --epoch 1: the CRF negative log-likelihood path of the HIP training step through the CLI."""
    L = 12
    argv = ['--dataset', 'ATIS-BIO', '--method', 'decompose', '--independent', '2',
            '--automata_path', tree['paths']['IIID'], '--rank', '100', '--seed', '1', '--beta', '0.9',
            '--embed_dim', '16', '--normalize_automata', 'none', '--rand_constant', '0', '--use_crf', '1',
            '--update_nonlinear', 'tanh', '--bz', '9', '--seq_max_len', str(L), '--epoch', '1', '--lr', '0.01',
            '--train_portion', '1.0', '--data_dir', tree['paths']['data_dir'], '--model_dir', str(tmp_path)]
    results, stats, res_path = cli.main(argv)
    assert len(stats['train_step']) == 1 and stats['train_step'][0]['tokens'] > 0
    saved = cli.load_res(res_path)
    losses = [float(line.split('LOSS:')[1]) for line in saved['logger'].record if 'LOSS:' in line]
    assert len(losses) == 1 and np.isfinite(losses[0]) and losses[0] > 0


def test_decompose_cli_trains_the_shipped_configuration_shape(tree, tmp_path):
    """--farnn 2 --use_crf 1 (what the reference's example configurations use) for one epoch through the CLI."""
    L = 12
    argv = ['--dataset', 'ATIS-BIO', '--method', 'decompose', '--independent', '2',
            '--automata_path', tree['paths']['IIID'], '--rank', '100', '--seed', '1', '--beta', '0.9',
            '--embed_dim', '16', '--normalize_automata', 'none', '--rand_constant', '0', '--use_crf', '1',
            '--farnn', '2', '--update_nonlinear', 'tanh', '--bz', '9', '--seq_max_len', str(L), '--epoch', '2',
            '--lr', '0.005', '--train_portion', '1.0', '--data_dir', tree['paths']['data_dir'],
            '--model_dir', str(tmp_path)]
    results, stats, res_path = cli.main(argv)
    assert len(stats['train_step']) == 2
    saved = cli.load_res(res_path)
    losses = [float(line.split('LOSS:')[1]) for line in saved['logger'].record if 'LOSS:' in line]
    assert len(losses) == 2 and np.isfinite(losses).all() and losses[1] < losses[0]

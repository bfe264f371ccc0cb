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
    automaton = tree['automaton']

    # with beta=1 and (nearly) exact CP factors the decomposed tagger reproduces the i-FST tags
    def score_fn(t2i, s2i, x, lengths):
        T, _, W, O, _, fin, sta, _ = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i)
        return fo.onehot_ifst_scores(T, W, O, sta, fin, x, lengths)

    tok, ent = _expected(tree, 'test', L, score_fn)
    assert abs(results['test']['token-level'][3] - tok[3]) < 0.05
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
    automaton = tree['automaton']

    # beta=1 and (nearly) exact CP factors of both tensors: the onehot independent=1 tags
    def score_fn(t2i, s2i, x, lengths):
        T, _, W, Oten, _, fin, sta, _ = f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i)
        return fo.onehot_ind1_scores(T, W, Oten, sta, fin, x, lengths)

    tok, ent = _expected(tree, 'test', L, score_fn)
    assert abs(results['test']['token-level'][3] - tok[3]) < 0.05
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

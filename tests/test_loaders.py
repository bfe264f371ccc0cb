"""Host-side boundary: automaton dict -> tensors, against the reference's own loader outputs
(fixture loader_small, both '%' conventions), plus the synthetic generators and the decomposed
pickle loader."""
import json
import os
import pickle

import numpy as np
import pytest

from re2nn_seq_amd import synth
from re2nn_seq_amd.wfa import fsa_to_tensor as f2t
from util import GOLDEN, load_golden, ns


def _automaton():
    with open(os.path.join(GOLDEN, 'loader_small.json')) as f:
        meta = json.load(f)
    a = meta['automaton']
    automaton = {'states': set(a['states']), 'startstate': a['startstate'], 'finalstates': a['finalstates'],
                 'transitions': {int(fr): {int(to): set(e) for to, e in d.items()}
                                 for fr, d in a['transitions'].items()}}
    return automaton, meta['t2i'], meta['s2i']


@pytest.mark.parametrize('dataset', ['MITR-BIO', 'ATIS-BIO'])
def test_dfa_to_tensor_variants_match_reference(dataset, capsys):
    automaton, t2i, s2i = _automaton()
    g = load_golden('loader_small')
    r = f2t.dfa_to_tensor_slot_new_wildcard(automaton, t2i, s2i, dataset=dataset)
    assert np.array_equal(r[0], g[dataset + '.new.T4']) and np.array_equal(r[2], g[dataset + '.new.W4'])
    assert np.array_equal(r[3], g[dataset + '.new.WW'])
    assert np.array_equal(r[4], g[dataset + '.new.final']) and np.array_equal(r[5], g[dataset + '.new.start'])
    assert sorted(r[6]) == sorted(g[dataset + '.new.language'].tolist())
    assert r[0].dtype == np.float64
    r = f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i, dataset=dataset)
    assert np.array_equal(r[0], g[dataset + '.ind.T']) and np.array_equal(r[2], g[dataset + '.ind.W'])
    assert np.array_equal(r[3], g[dataset + '.ind.Oten']) and r[4] is None
    r = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i, dataset=dataset)
    assert np.array_equal(r[0], g[dataset + '.single.T']) and np.array_equal(r[2], g[dataset + '.single.W'])
    assert np.array_equal(r[3], g[dataset + '.single.O']) and np.array_equal(r[4], g[dataset + '.single.Ow'])
    assert np.array_equal(r[5], g[dataset + '.single.final']) and np.array_equal(r[6], g[dataset + '.single.start'])


def test_default_dataset_kwarg_is_mitr_quirk():
    """The onehot driver never passes `dataset`, so '%' means the integers 0..24 only."""
    automaton, t2i, s2i = _automaton()
    a = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i)
    b = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i, dataset='MITR-BIO')
    assert np.array_equal(a[0], b[0])
    assert f2t.is_small_pos_number('24') and not f2t.is_small_pos_number('25')
    assert f2t.is_number('2021') and f2t.is_number('9.5') and not f2t.is_number('w3')


def test_oov_rule_word_keeps_its_label(capsys):
    """A rule word outside the vocabulary: the reference writes the edge's label before the vocabulary lookup
    (fsa_to_tensor.py:515, :586), so output_mat / output_tensor keep it and only the language tensor skips the
    edge; the 4-D layout drops it.  Fixture captured from the reference loaders (make_golden_oov.py)."""
    edges = [(0, 1, 'notaword<:>b-e1'), (2, 3, 'alsomissing<:>i-e2'), (1, 1, 'nope<:>oo')]   # = make_golden_oov.OOV_EDGES
    automaton, t2i, s2i = _automaton()
    for fr, to, lab in edges:
        automaton['transitions'].setdefault(fr, {}).setdefault(to, set()).add(lab)
    g = load_golden('loader_oov')
    r = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i)
    assert 'OOV word: notaword in rule' in capsys.readouterr().out
    assert np.array_equal(r[0], g['single.T']) and np.array_equal(r[2], g['single.W'])
    assert np.array_equal(r[3], g['single.O'])
    base = f2t.dfa_to_tensor_slot_single_wildcard(_automaton()[0], t2i, s2i)
    assert not np.array_equal(base[3], g['single.O'])            # the fixture discriminates: labels were added
    assert np.array_equal(base[0], g['single.T'])                # ... and no language-tensor entry was
    r = f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i)
    assert np.array_equal(r[0], g['ind.T']) and np.array_equal(r[2], g['ind.W']) and np.array_equal(r[3], g['ind.Oten'])
    a4 = {**automaton, 'transitions': {f: {t: {e for e in es if e != 'nope<:>oo'} for t, es in d.items()}
                                       for f, d in automaton['transitions'].items()}}
    r = f2t.dfa_to_tensor_slot_new_wildcard(a4, t2i, s2i)
    assert np.array_equal(r[0], g['new.T4']) and np.array_equal(r[2], g['new.W4'])
    # the device-side builders get the same label-only entries (word = -2)
    w, fr_, to_, lab, _, _, _ = f2t.dfa_to_edges_slot_single_wildcard(automaton, t2i, s2i)
    assert (w == -2).sum() >= len(edges)
    O = np.zeros_like(g['single.O'])
    O[lab[lab >= 0], to_[lab >= 0]] = 1
    assert np.array_equal(O, g['single.O'])


def test_synthetic_automaton_has_ifst_property():
    dset, automaton, rules = synth.make_dataset(60, 4, 25, seed=3)
    incoming = {}
    for fr, to in automaton['transitions'].items():
        for t, edges in to.items():
            for e in edges:
                incoming.setdefault(t, set()).add(e.split('<:>')[1])
    assert all(len(tags) == 1 for tags in incoming.values())        # one label per destination state
    assert 0 in automaton['finalstates'] and automaton['startstate'] == [0]
    assert len(automaton['states']) == 25 and len(rules) > 0


def test_exact_cp_factors_reconstruct_the_language_tensor():
    dset, automaton, _ = synth.make_dataset(40, 3, 12, seed=5)
    T = f2t.dfa_to_tensor_slot_single_wildcard(automaton, dset['t2i'], dset['s2i'])[0]
    V, S1, S2 = synth.exact_cp_factors(T)
    assert np.array_equal(np.einsum('vr,sr,jr->vsj', V, S1, S2), T)


def test_decomposed_pickle_loader(tmp_path):
    """IIID pickle -> get_init_params_seq_independent_single (reference init_params.py:221-320)."""
    from re2nn_seq_amd.init_params import get_init_params_seq_independent_single
    rng = np.random.RandomState(0)
    dset, automaton, _ = synth.make_dataset(40, 3, 12, seed=5)
    t2i, s2i = dset['t2i'], dset['s2i']
    n_pairs = 40
    blob = synth.make_iiid_pickle_dict(automaton, t2i, s2i, ranks=[100], rng=rng)
    ddir = tmp_path / 'ATIS-BIO'
    ddir.mkdir()
    apath = ddir / 'IIID.automata.synthetic.pkl'
    with open(apath, 'wb') as f:
        pickle.dump(blob, f)
    D = 16
    with open(ddir / 'glove.{}.emb'.format(D), 'wb') as f:
        pickle.dump(rng.randn(len(t2i), D), f)
    a = ns(dataset='ATIS-BIO', embed_type='glove', embed_dim=D, random_embed=0, automata_path=str(apath),
           seed=1, rank=100, normalize_automata='l2-rank', use_bert=0)
    out = get_init_params_seq_independent_single(a, s2i, t2i, data_dir=str(tmp_path) + '/')
    V_ext, S1, S2, E_ext, W, Ow, fin, sta, pri, Cout, bert = out
    assert V_ext.shape == (len(t2i) + 1, 100) and np.all(V_ext[-1] == 0)        # pad row appended
    assert E_ext.shape == (len(t2i) + 1, D) and np.all(E_ext[-1] == 0)
    assert Cout.shape == (len(s2i) + 1, 12)                                     # CE1 -> entry [2]
    assert fin.sum() == len(automaton['finalstates']) and sta[0] == 1
    assert pri.shape == (len(s2i), len(s2i)) and (pri == -1).sum() == 3          # ATIS: i-x <- b-x
    # l2-rank normalisation equalises the average column norms of the three factors (:285-297)
    from re2nn_seq_amd.utils import get_average
    avgs = [get_average(m, 'l2-rank') for m in (V_ext, S1, S2)]
    nz = avgs[0] > 0
    np.testing.assert_allclose(avgs[0][nz], avgs[1][nz], rtol=1e-9)
    np.testing.assert_allclose(avgs[1][nz], avgs[2][nz], rtol=1e-9)
    assert bert is None and n_pairs > 0


def test_decomposed_independent1_pickle_loader(tmp_path):
    """IID pickle -> get_init_params_seq_independent (reference init_params.py:123-218)."""
    from re2nn_seq_amd.init_params import get_init_params_seq_independent
    from re2nn_seq_amd.utils import get_average
    rng = np.random.RandomState(0)
    dset, automaton, _ = synth.make_dataset(40, 3, 12, seed=5)
    t2i, s2i = dset['t2i'], dset['s2i']
    blob = synth.make_iid_pickle_dict(automaton, t2i, s2i, ranks=[100], output_ranks=[70], rng=rng)
    # the exact factors reproduce the C+1 output tensor of the reference's CE1 layout
    Oten = f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i)[3]
    o = blob[1][2][70]
    np.testing.assert_allclose(np.einsum('cq,iq,jq->cij', o['C_output'], o['S1_output'], o['S2_output']),
                               Oten, atol=0.2)
    assert o['wildcard_output'] is None and blob[1][1][70]['wildcard_output'].shape == Oten.shape[1:]
    ddir = tmp_path / 'ATIS-BIO'
    ddir.mkdir()
    apath = ddir / 'IID.automata.synthetic.pkl'
    with open(apath, 'wb') as f:
        pickle.dump(blob, f)
    D = 16
    with open(ddir / 'glove.{}.emb'.format(D), 'wb') as f:
        pickle.dump(rng.randn(len(t2i), D), f)
    for loss, C in (('CE1', len(s2i) + 1), ('CE', len(s2i))):
        a = ns(dataset='ATIS-BIO', embed_type='glove', embed_dim=D, random_embed=0, automata_path=str(apath),
               seed=1, rank=100, rank_wildcard=70, normalize_automata='l2-rank', use_bert=0,
               local_loss_func=loss)
        out = get_init_params_seq_independent(a, s2i, t2i, data_dir=str(tmp_path) + '/')
        V_ext, S1, S2, E_ext, W, Wo, fin, sta, pri, Cout, S1o, S2o = out
        assert V_ext.shape == (len(t2i) + 1, 100) and np.all(V_ext[-1] == 0)
        assert E_ext.shape == (len(t2i) + 1, D) and np.all(E_ext[-1] == 0)
        assert Cout.shape == (C, 70) and S1o.shape == (12, 70) and S2o.shape == (12, 70)
        assert (Wo is None) == (loss == 'CE1')
        assert fin.sum() == len(automaton['finalstates']) and sta[0] == 1
        for trip in ((V_ext, S1, S2), (Cout, S1o, S2o)):      # both triples are normalised (:191-215)
            avgs = [get_average(m, 'l2-rank') for m in trip]
            nz = avgs[0] > 0
            np.testing.assert_allclose(avgs[0][nz], avgs[1][nz], rtol=1e-9)
            np.testing.assert_allclose(avgs[1][nz], avgs[2][nz], rtol=1e-9)


def test_decomposed_independent0_pickle_loader(tmp_path):
    """D pickle -> get_init_params_seq (reference init_params.py:10-121)."""
    from re2nn_seq_amd.init_params import get_init_params_seq
    from re2nn_seq_amd.utils import get_average
    rng = np.random.RandomState(0)
    dset, automaton, _ = synth.make_dataset(40, 3, 12, seed=5)
    t2i, s2i = dset['t2i'], dset['s2i']
    blob = synth.make_d_pickle_dict(automaton, t2i, s2i, ranks=[100], wildcard_ranks=[70], rng=rng)
    T4, _, W4, WW, _, _, _ = f2t.dfa_to_tensor_slot_new_wildcard(automaton, t2i, s2i)
    d = blob[1][0][100]
    np.testing.assert_allclose(np.einsum('vr,cr,sr,jr->vcsj', d['V'], d['C'], d['S1'], d['S2']), T4, atol=0.2)
    ddir = tmp_path / 'ATIS-BIO'
    ddir.mkdir()
    apath = ddir / 'D.automata.synthetic.pkl'
    with open(apath, 'wb') as f:
        pickle.dump(blob, f)
    D = 16
    with open(ddir / 'glove.{}.emb'.format(D), 'wb') as f:
        pickle.dump(rng.randn(len(t2i), D), f)
    a = ns(dataset='ATIS-BIO', embed_type='glove', embed_dim=D, random_embed=0, automata_path=str(apath),
           seed=1, rank=100, rank_wildcard=70, normalize_automata='l2-rank', use_bert=0, local_loss_func='CE1')
    out = get_init_params_seq(a, s2i, data_dir=str(tmp_path) + '/')
    V_ext, Cemb, S1, S2, E_ext, Wt, WWt, fin, sta, pri, Cw, S1w, S2w = out
    assert V_ext.shape == (len(t2i) + 1, 100) and np.all(V_ext[-1] == 0)
    assert Cemb.shape == (len(s2i) + 1, 100) and Cw.shape == (len(s2i) + 1, 70)
    assert Wt.shape == W4.shape and WWt.shape == WW.shape
    assert S1w.shape == (12, 70) and S2w.shape == (12, 70)
    avgs = [get_average(m, 'l2-rank') for m in (V_ext, Cemb, S1, S2)]      # 4th-root normalisation (:91-107)
    nz = avgs[0] > 0
    for other in avgs[1:]:
        np.testing.assert_allclose(avgs[0][nz], other[nz], rtol=1e-9)
    # values beyond +-100 are clipped to +-1 by this loader only (:50-65)
    blob[1][0][100]['S1'][0, 0] = 1e4
    with open(apath, 'wb') as f:
        pickle.dump(blob, f)
    a.normalize_automata = 'none'
    assert get_init_params_seq(a, s2i, data_dir=str(tmp_path) + '/')[2][0, 0] == 1


def test_edge_list_reproduces_the_dense_ifst_tensors():
    """dfa_to_edges_slot_single_wildcard: scattering its entries gives exactly the tensors of
    dfa_to_tensor_slot_single_wildcard (what farnn_onehot_ifst_create_from_edges does on the device)."""
    for ds in ('MITR-BIO', 'ATIS-BIO'):
        dset, automaton, _ = synth.make_dataset(40, 3, 12, seed=5)
        t2i = dict(dset['t2i']); t2i['<pad>'] = len(t2i)
        s2i = dset['s2i']
        T, _, W, O, _, fin, sta, _ = f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i, dataset=ds)
        word, frm, to, label, fin2, sta2, _ = f2t.dfa_to_edges_slot_single_wildcard(automaton, t2i, s2i, dataset=ds)
        assert word.dtype == np.int32 and len(word) == len(frm) == len(to) == len(label)
        T2, W2, O2 = np.zeros_like(T), np.zeros_like(W), np.zeros_like(O)
        for w, f, t, l in zip(word, frm, to, label):
            if w >= 0:
                T2[w, f, t] = 1
            elif w == -1:
                W2[f, t] = 1
            O2[l, t] = 1
        assert np.array_equal(T2, T) and np.array_equal(W2, W) and np.array_equal(O2, O)
        assert np.array_equal(fin2, fin) and np.array_equal(sta2, sta)


def test_edge_lists_reproduce_the_4d_and_independent1_tensors():
    dset, automaton, _ = synth.make_dataset(40, 3, 12, seed=5)
    t2i = dict(dset['t2i']); t2i['<pad>'] = len(t2i)
    s2i = dset['s2i']
    T4, _, W4, _, fin, sta, _ = f2t.dfa_to_tensor_slot_new_wildcard(automaton, t2i, s2i)
    word, frm, to, label, fin2, sta2, _ = f2t.dfa_to_edges_slot_new_wildcard(automaton, t2i, s2i)
    A, Wd = np.zeros_like(T4), np.zeros_like(W4)
    lang, wild = word >= 0, word == -1
    A[word[lang], label[lang], frm[lang], to[lang]] = 1
    Wd[label[wild], frm[wild], to[wild]] = 1
    assert np.array_equal(A, T4) and np.array_equal(Wd, W4) and np.array_equal(fin, fin2) and np.array_equal(sta, sta2)
    T, _, W, Oten, _, _, _, _ = f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i)
    word, frm, to, label, _, _, _ = f2t.dfa_to_edges_slot_independent_wildcard(automaton, t2i, s2i)
    T2, W2, O2 = np.zeros_like(T), np.zeros_like(W), np.zeros_like(Oten)
    lang, wild = word >= 0, word == -1
    T2[word[lang], frm[lang], to[lang]] = 1
    W2[frm[wild], to[wild]] = 1
    O2[label, frm, to] = 1
    assert np.array_equal(T2, T) and np.array_equal(W2, W) and np.array_equal(O2, Oten)

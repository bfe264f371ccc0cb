"""synth.planted_rule_ifst / planted_rule_batch (round 6): the rule automaton behind tests/golden/bench_decomp_exact.npz.  Its claims,
checked on a small instance and at SNIPS-BIO size: the i-FST property (one incoming label per state: fsa_to_tensor.py:586), at most
`max_pairs` (from, to) pairs, disjoint word sets among a state's outgoing pairs, and that the rank-`max_pairs` factors reproduce the
language tensor EXACTLY (T[w,s,j] = sum_r Vgen[w,r] S1[s,r] S2[j,r]: the layout of decompose_automata.py:373-431)."""
import numpy as np
import pytest

from re2nn_seq_amd import synth
from oracle import farnn_oracle as fo


@pytest.mark.parametrize('V,S,C,P', [(300, 24, 9, 60), (11000, 104, 73, 250)])
def test_planted_automaton_is_an_ifst_with_an_exact_cp_form(V, S, C, P):
    A = synth.planted_rule_ifst(seed=5, V=V, S=S, C=C, max_pairs=P)
    assert len(A['pairs']) <= P and A['Vgen'].shape == (V, P) and A['S1'].shape == (S, P)
    # every word edge lands on a state whose one label is state_label[to]; wildcard states carry `oo`
    lab = A['state_label']
    assert (lab[[0, S - 1]] == C - 1).all() and ((lab >= 0) & (lab < C)).all()
    assert (A['O'].sum(0) == 1).all() and (A['O'][lab, np.arange(S)] == 1).all()
    # disjoint word sets among the outgoing pairs of a state (the sink re-uses a rule's first set: its own outgoing pairs are disjoint too)
    for frm in set(a for a, _ in A['pairs']):
        ws = np.concatenate([A['pair_words'][r] for r, (a, _) in enumerate(A['pairs']) if a == frm])
        assert len(ws) == len(np.unique(ws)), frm
    assert (A['word'] < V - 1).all()                       # the pad row stays empty
    # exact CP form, on the words the automaton uses (dense at SNIPS size: 476 MB -- a sample of words there)
    words = np.unique(A['word'][A['word'] >= 0])
    if V > 1000:
        words = words[:: max(1, len(words) // 200)]
    m = np.isin(A['word'], words)
    T = np.zeros((V, S, S), np.float32) if V <= 1000 else None
    for w in words:
        Tw = np.zeros((S, S), np.float32)
        sel = A['word'] == w
        Tw[A['frm'][sel], A['to'][sel]] = 1.0
        cp = np.einsum('r,sr,jr->sj', A['Vgen'][w], A['S1'], A['S2'])
        assert np.array_equal(cp, Tw), w
    assert m.any()


def test_planted_batch_fires_rules_and_decomposed_equals_onehot_oracle():
    """small instance: the oracle's decomposed scores on the exact factors (update_nonlinear = none) equal its onehot scores bit for bit,
    and the planted batch decodes to many tags"""
    V, S, C, P = 300, 24, 9, 60
    A = synth.planted_rule_ifst(seed=5, V=V, S=S, C=C, max_pairs=P)
    x, lengths = synth.planted_rule_batch(A, 32, 24, seed=3, max_rules=4)      # (the number of accepting paths doubles with every planted rule)
    T = synth.dense_from_edges(A)
    so = fo.onehot_ifst_scores(T, A['W'], A['O'], A['h0'], A['hT'], x, lengths)
    q = {'Vgen': A['Vgen'], 'S1': A['S1'], 'S2': A['S2'], 'W': A['W'], 'Cout': A['O'], 'h0': A['h0'], 'hT': A['hT'], 'farnn': 0,
         'nl': fo.NL_NONE, 'semiring': 0, 'sig_k': 5}
    sd = fo.decomp_ifst_scores(q, x, lengths)
    mask = np.arange(sd.shape[1])[None, :] < lengths[:, None]
    assert float(np.abs(so).max()) < 2.0 ** 22 and np.array_equal(so[:, :sd.shape[1]][mask], sd[mask])
    tags = fo.forward_local_tags(sd, lengths, 0.5, 0)
    u, c = np.unique(tags, return_counts=True)
    assert len(u) >= 5 and c.max() <= 0.8 * c.sum()

"""Automaton dict -> dense numpy tensors: the ".pkl loader" side of the tagging path.

Boundary restatement of the reference's ``src_seq/wfa/fsa_to_tensor.py:398-615``
(``dfa_to_tensor_slot_{new,independent,single}_wildcard``).  The three public functions
keep the reference's names, argument order and return tuples so that the drivers read the
same; internally they share one edge walker that reports every edge as
``(word ids, label column, from index, to index, is_wildcard)``.

Automaton dict schema (writer: reference ``create_dataset_automata.py:109-141``):
    states: set[int]   startstate: list[int]   finalstates: list[int]
    transitions: {from_state: {to_state: set['word<:>tag']}}
Edge words: a literal word, ``$`` (any word), ``%`` (a number), ``&`` (punctuation);
tag ``oo`` means "any label" and maps to column ``len(slot2idx)``.

Quirks preserved on purpose (SURVEY.md 8b "loader quirks"):
  * state index = enumeration order of ``list(automata['states'])`` (ref :558-561);
  * ``dataset`` defaults to ``'MITR-BIO'``, for which ``%`` matches only the integers
    0..24 (ref :47-56, :63-66) -- the onehot driver never passes the kwarg;
  * the label of an edge is attached to its DESTINATION state in the i-FST layout (ref :586);
  * ``final_vector`` / ``start_vector`` are indexed with the raw state ids (ref :606-610);
  * a rule word missing from the vocabulary is reported and its language-tensor entry skipped
    (ref :603); its LABEL is still written to output_mat / output_tensor (ref :515, :586), and the 4-D
    layout, which has no separate label array, drops the edge entirely (ref :463-465).
All outputs are float64, like the reference's.
"""
import numpy as np

PUNCTUATIONS = frozenset(
    [',', '，', ':', '：', '!', '！', '《', '》', '。', '；', '.',
     '(', ')', '（', '）', '|', '?', '"'])

_NUMBER_RULE = {            # dataset -> "%" predicate (ref get_num_punct :62-85)
    'MITR-BIO': 'small',
    'MITM-E-BIO': 'any', 'ATIS-BIO': 'any', 'ATIS-ZH-BIO': 'any', 'SNIPS-BIO': 'any',
}


def is_punct(token):
    return token in PUNCTUATIONS


def is_number(token):
    """ref :58-59"""
    return token.replace('.', '', 1).isdigit()


def is_small_pos_number(token):
    """ref :47-56: an integer literal (one '.' tolerated) in [0, 25)."""
    try:
        value = int(token.replace('.', '', 1))
    except (ValueError, TypeError):
        return False
    return 0 <= value < 25


def get_num_punct(word2idx, dataset):
    """ref :62-85: the word->id maps matched by '%' and '&'."""
    if dataset not in _NUMBER_RULE:
        raise NotImplementedError(dataset)
    pred = is_small_pos_number if _NUMBER_RULE[dataset] == 'small' else is_number
    numbers = {w: i for w, i in word2idx.items() if pred(w)}
    puncts = {w: i for w, i in word2idx.items() if is_punct(w)}
    return numbers, puncts


def _walk_edges(automata, word2idx, slot2idx, dataset, strict_oo):
    """Yield (word_id_list | None, label_col, from_idx, to_idx, words_matched) per edge.

    word_id_list is None for a wildcard ('$') edge."""
    state2idx = {s: k for k, s in enumerate(list(automata['states']))}
    numbers, puncts = get_num_punct(word2idx, dataset)
    n_slots = len(slot2idx)
    edges = []
    for fr_state, fanout in sorted(automata['transitions'].items()):
        for to_state, labels in sorted(fanout.items()):
            for edge in labels:
                word, slot = edge.split('<:>')
                if slot == 'oo':
                    col = n_slots
                    if strict_oo:
                        assert word == '$'
                else:
                    col = slot2idx[slot]
                fi, ti = state2idx[fr_state], state2idx[to_state]
                if word == '&':
                    edges.append((list(puncts.values()), col, fi, ti, list(puncts.keys())))
                elif word == '%':
                    edges.append((list(numbers.values()), col, fi, ti, list(numbers.keys())))
                elif word == '$':
                    edges.append((None, col, fi, ti, []))
                elif word in word2idx:
                    edges.append(([word2idx[word]], col, fi, ti, [word]))
                else:
                    # the reference writes the edge's label BEFORE it looks the word up (:120, :515, :586), so an
                    # out-of-vocabulary rule word still labels output_mat / output_tensor; only the language
                    # tensor entry is skipped.  An empty word list is exactly that: label only.
                    print('OOV word: {} in rule'.format(word))
                    edges.append(([], col, fi, ti, []))
    return state2idx, edges


def _start_final(automata, n_states):
    final_vector = np.zeros(n_states)
    final_vector[automata['finalstates']] = 1
    start_vector = np.zeros(n_states)
    start_vector[automata['startstate']] = 1
    return final_vector, start_vector


def dfa_to_tensor_slot_new_wildcard(automata, word2idx, slot2idx, dataset='MITR-BIO'):
    """FST layout (independent=0): T4[V,C,S,S], W4[C,S,S]; ref :398-474."""
    state2idx, edges = _walk_edges(automata, word2idx, slot2idx, dataset, strict_oo=True)
    S, C, V = len(automata['states']), len(slot2idx) + 1, len(word2idx)
    language_tensor = np.zeros((V, C, S, S))
    wildcard_tensor = np.zeros((C, S, S))
    wildcard_wildcard_tensor = np.zeros((S, S))
    language = set()
    for wids, col, fi, ti, words in edges:
        if wids is None:
            wildcard_tensor[col, fi, ti] = 1
        else:
            language_tensor[wids, col, fi, ti] = 1
            language.update(words)
    final_vector, start_vector = _start_final(automata, S)
    print("LANGUAGE SET SIZE: {}".format(len(language)))
    return (language_tensor, state2idx, wildcard_tensor, wildcard_wildcard_tensor,
            final_vector, start_vector, sorted(language))


def dfa_to_tensor_slot_independent_wildcard(automata, word2idx, slot2idx, dataset='MITR-BIO'):
    """independent=1 layout: T[V,S,S], W[S,S], Oten[C,S,S]; ref :477-543."""
    state2idx, edges = _walk_edges(automata, word2idx, slot2idx, dataset, strict_oo=False)
    S, C, V = len(automata['states']), len(slot2idx) + 1, len(word2idx)
    language_tensor = np.zeros((V, S, S))
    language_wildcard_mat = np.zeros((S, S))
    output_tensor = np.zeros((C, S, S))
    language = set()
    for wids, col, fi, ti, words in edges:
        output_tensor[col, fi, ti] = 1
        if wids is None:
            language_wildcard_mat[fi, ti] = 1
        else:
            language_tensor[wids, fi, ti] = 1
            language.update(words)
    final_vector, start_vector = _start_final(automata, S)
    print("LANGUAGE SET SIZE: {}".format(len(language)))
    return (language_tensor, state2idx, language_wildcard_mat, output_tensor, None,
            final_vector, start_vector, sorted(language))


def dfa_to_tensor_slot_single_wildcard(automata, word2idx, slot2idx, dataset='MITR-BIO'):
    """i-FST layout (independent=2): T[V,S,S], W[S,S], O[C,S]; ref :546-615."""
    state2idx, edges = _walk_edges(automata, word2idx, slot2idx, dataset, strict_oo=False)
    S, C, V = len(automata['states']), len(slot2idx) + 1, len(word2idx)
    language_tensor = np.zeros((V, S, S))
    language_wildcard_mat = np.zeros((S, S))
    output_mat = np.zeros((C, S))
    output_wildcard_vector = np.zeros(S)
    language = set()
    for wids, col, fi, ti, words in edges:
        output_mat[col, ti] = 1
        if wids is None:
            language_wildcard_mat[fi, ti] = 1
        else:
            language_tensor[wids, fi, ti] = 1
            language.update(words)
    final_vector, start_vector = _start_final(automata, S)
    print("LANGUAGE SET SIZE: {}".format(len(language)))
    return (language_tensor, state2idx, language_wildcard_mat, output_mat,
            output_wildcard_vector, final_vector, start_vector, sorted(language))


def _edge_arrays(automata, word2idx, slot2idx, dataset, strict_oo):
    state2idx, edges = _walk_edges(automata, word2idx, slot2idx, dataset, strict_oo=strict_oo)
    word, frm, to, label = [], [], [], []
    for wids, col, fi, ti, _ in edges:
        for w in ([-1] if wids is None else wids):
            word.append(w); frm.append(fi); to.append(ti); label.append(col)
        if wids is not None and not wids:         # an empty class still carries its label
            word.append(-2); frm.append(fi); to.append(ti); label.append(col)
    final_vector, start_vector = _start_final(automata, len(automata['states']))
    as32 = lambda a: np.asarray(a, dtype=np.int32)      # noqa: E731
    return as32(word), as32(frm), as32(to), as32(label), final_vector, start_vector, state2idx


def dfa_to_edges_slot_single_wildcard(automata, word2idx, slot2idx, dataset='MITR-BIO'):
    """The edges dfa_to_tensor_slot_single_wildcard (ref :546-615) would write, as flat int32 arrays
    for the device-side builder (include/farnn.h: farnn_edge_list): one entry per (word, from, to)
    with the class edges '&' / '%' expanded, word = -1 for '$' edges (-2: label only, a class with no
    member in the vocabulary), and the label column of the edge (= of its destination state).
    Returns (word, frm, to, label, final_vector, start_vector, state2idx)."""
    return _edge_arrays(automata, word2idx, slot2idx, dataset, strict_oo=False)


def dfa_to_edges_slot_independent_wildcard(automata, word2idx, slot2idx, dataset='MITR-BIO'):
    """Edge arrays of dfa_to_tensor_slot_independent_wildcard (ref :477-543): same entries, the label
    goes to Oten[label, from, to]."""
    return _edge_arrays(automata, word2idx, slot2idx, dataset, strict_oo=False)


def dfa_to_edges_slot_new_wildcard(automata, word2idx, slot2idx, dataset='MITR-BIO'):
    """Edge arrays of dfa_to_tensor_slot_new_wildcard (ref :398-474, 4-D FST): T4[word, label, from, to]
    and W4[label, from, to]; 'oo' must sit on '$' edges (the reference asserts it)."""
    return _edge_arrays(automata, word2idx, slot2idx, dataset, strict_oo=True)

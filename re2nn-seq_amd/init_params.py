"""Decomposed-automaton pickle -> initial parameters (reference src_seq/init_params.py:221-320,
``get_init_params_seq_independent_single``: the ``--independent 2`` loader; :123-218,
``get_init_params_seq_independent``: the ``--independent 1`` loader; :10-121, ``get_init_params_seq``:
the ``--independent 0`` loader -- schemas in their docstrings).

Pickle schema (writer: reference wfa/decompose_automata.py:373-431):
    {'automata': dict,
     seed: [ {rank: {'V':[V,R], 'S1':[S,R], 'S2':[S,R], 'wildcard_mat':[S,S]}},
             {'output_mat':[C,S],   'output_wildcard_vector':[S]},      # non-CE1
             {'output_mat':[C+1,S], 'output_wildcard_vector':[S]} ]}    # CE1
Quirks kept: `[args.rank]` indexes only the factor dict (:239-240); CE1 picks entry [2]
(:234-235); a zero pad row is appended to V_embed and to the word embeddings (:280-281);
start/final vectors are indexed with raw state ids.  numpy-2 note: the reference's `np.float`
is spelled `np.float64` here.
"""
import os

import numpy as np

from .create_logic_mat_bias import create_mat_priority
from .data import load_fasttext_embed, load_glove_embed
from .utils import get_average, load_pkl, xavier_normal


def get_init_params_seq_independent_single(args, s2i, t2i, data_dir='../data/'):
    print("Start getting initial decompsoed parameters V C S1 S2...")
    dpath = os.path.join(data_dir, args.dataset)
    loader = load_glove_embed if args.embed_type == 'glove' else load_fasttext_embed
    pretrained_embed = loader(dpath, args.embed_dim)
    if args.random_embed:
        pretrained_embed = np.random.random(pretrained_embed.shape)

    automata_dicts = load_pkl(args.automata_path)
    print("Loading automata: {}".format(args.automata_path))
    per_seed = automata_dicts[args.seed]
    factor_dicts = per_seed[0][args.rank]
    factor_output_dicts = per_seed[2] if args.local_loss_func == 'CE1' else per_seed[1]
    automata = automata_dicts['automata']

    V_embed, S1, S2 = factor_dicts['V'], factor_dicts['S1'], factor_dicts['S2']
    wildcard_mat = factor_dicts['wildcard_mat']
    C_output_mat = factor_output_dicts['output_mat']
    wildcard_output_vector = factor_output_dicts['output_wildcard_vector']

    corrupt = 1e5          # the reference only reports these counts (:249-262)
    print('Invalid Positive Values: {}'.format(
        int(np.sum(V_embed > corrupt) + np.sum(S1 > corrupt) + np.sum(S2 > corrupt))))
    print('Invalid Negative Values: {}'.format(
        int(np.sum(V_embed < -corrupt) + np.sum(S1 < -corrupt) + np.sum(S2 < -corrupt))))

    n_state, rank = S1.shape
    final_vector = np.zeros(n_state)
    final_vector[automata['finalstates']] = 1
    start_vector = np.zeros(n_state)
    start_vector[automata['startstate']] = 1
    print("DFA states: {}".format(n_state))

    pretrain_embed_extend = np.append(pretrained_embed, np.zeros((1, args.embed_dim), dtype=np.float64), axis=0)
    V_embed_extend = np.append(V_embed, np.zeros((1, rank), dtype=np.float64), axis=0)
    priority_mat = create_mat_priority(s2i, args)

    if args.normalize_automata != 'none':       # (:285-297)
        print('Normalize automata decomposed parameters...')
        v_avg = get_average(V_embed_extend, args.normalize_automata)
        s1_avg = get_average(S1, args.normalize_automata)
        s2_avg = get_average(S2, args.normalize_automata)
        factor = np.float_power(v_avg * s1_avg * s2_avg, 1 / 3)
        S1 = S1 * (factor / s1_avg)
        S2 = S2 * (factor / s2_avg)
        V_embed_extend = V_embed_extend * (factor / v_avg)

    if args.random == 1:                         # (:299-308)
        V_embed_extend, S1, S2 = xavier_normal(V_embed_extend), xavier_normal(S1), xavier_normal(S2)
        wildcard_mat = xavier_normal(wildcard_mat)
        wildcard_output_vector = xavier_normal(wildcard_output_vector)
        final_vector, start_vector = xavier_normal(final_vector), xavier_normal(start_vector)
        C_output_mat = xavier_normal(C_output_mat)
        assert args.use_priority == 0

    if bool(getattr(args, 'use_bert', 0)):
        raise NotImplementedError('the BERT front-end is outside the forward tagging path (SURVEY.md 2, row 23)')
    return (V_embed_extend, S1, S2, pretrain_embed_extend, wildcard_mat, wildcard_output_vector,
            final_vector, start_vector, priority_mat, C_output_mat, None)


def get_init_params_seq_independent(args, s2i, t2i, data_dir='../data/'):
    """``--independent 1`` loader (reference init_params.py:123-218).  Pickle schema (writer:
    decompose_automata.py:148-300):
        seed: [ {rank: {'V','S1','S2','wildcard_mat'}},
                {rank_wildcard: {'C_output'[C,RO],'S1_output','S2_output'[S,RO],'wildcard_output'[S,S]}},
                {rank_wildcard: {'C_output'[C+1,RO], ..., 'wildcard_output': None}} ]        # CE1
    Quirks kept: the output factors are indexed by ``args.rank_wildcard`` (:144); BOTH factor
    triples are normalised (:191-215); no ``args.random`` branch here (the model re-initialises)."""
    print("Start getting initial decompsoed parameters V C S1 S2...")
    dpath = os.path.join(data_dir, args.dataset)
    loader = load_glove_embed if args.embed_type == 'glove' else load_fasttext_embed
    pretrained_embed = loader(dpath, args.embed_dim)
    if args.random_embed:
        pretrained_embed = np.random.random(pretrained_embed.shape)

    automata_dicts = load_pkl(args.automata_path)
    print("Loading automata: {}".format(args.automata_path))
    per_seed = automata_dicts[args.seed]
    factor_dicts = per_seed[0][args.rank]
    factor_output_dicts = (per_seed[2] if args.local_loss_func == 'CE1' else per_seed[1])[args.rank_wildcard]
    automata = automata_dicts['automata']

    V_embed, S1, S2 = factor_dicts['V'], factor_dicts['S1'], factor_dicts['S2']
    wildcard_mat = factor_dicts['wildcard_mat']
    C_output, S1_output, S2_output = (factor_output_dicts[k] for k in ('C_output', 'S1_output', 'S2_output'))
    wildcard_output = factor_output_dicts['wildcard_output']

    corrupt = 1000         # only reported (:155-170)
    print('Invalid Positive Values: {}'.format(
        int(np.sum(V_embed > corrupt) + np.sum(S1 > corrupt) + np.sum(S2 > corrupt))))
    print('Invalid Negative Values: {}'.format(
        int(np.sum(V_embed < -corrupt) + np.sum(S1 < -corrupt) + np.sum(S2 < -corrupt))))

    n_state, rank = S1.shape
    final_vector = np.zeros(n_state)
    final_vector[automata['finalstates']] = 1
    start_vector = np.zeros(n_state)
    start_vector[automata['startstate']] = 1
    print("DFA states: {}".format(n_state))

    pretrain_embed_extend = np.append(pretrained_embed, np.zeros((1, args.embed_dim), dtype=np.float64), axis=0)
    V_embed_extend = np.append(V_embed, np.zeros((1, rank), dtype=np.float64), axis=0)
    priority_mat = create_mat_priority(s2i, args)

    if args.normalize_automata != 'none':       # (:191-215)
        print('Normalize automata decomposed parameters...')
        v_avg = get_average(V_embed_extend, args.normalize_automata)
        s1_avg = get_average(S1, args.normalize_automata)
        s2_avg = get_average(S2, args.normalize_automata)
        factor = np.float_power(v_avg * s1_avg * s2_avg, 1 / 3)
        S1 = S1 * (factor / s1_avg)
        S2 = S2 * (factor / s2_avg)
        V_embed_extend = V_embed_extend * (factor / v_avg)
        c_avg = get_average(C_output, args.normalize_automata)
        s1o_avg = get_average(S1_output, args.normalize_automata)
        s2o_avg = get_average(S2_output, args.normalize_automata)
        factor = np.float_power(c_avg * s1o_avg * s2o_avg, 1 / 3)
        C_output = C_output * (factor / c_avg)
        S1_output = S1_output * (factor / s1o_avg)
        S2_output = S2_output * (factor / s2o_avg)

    return (V_embed_extend, S1, S2, pretrain_embed_extend, wildcard_mat, wildcard_output,
            final_vector, start_vector, priority_mat, C_output, S1_output, S2_output)


def get_init_params_seq(args, s2i, data_dir='../data/'):
    """``--independent 0`` loader (reference init_params.py:10-121).  Pickle schema (writer:
    decompose_automata.py:30-146):
        seed: [ {rank: {'V','C','S1','S2','wildcard_tensor'[C,S,S],'wildcard_wildcard_tensor'[S,S]}},
                {rank_wildcard: {'C_wildcard'[C,RW],'S1_wildcard','S2_wildcard'[S,RW]}} ]
    Quirks kept: no CE1 variant exists (:28) -- under CE1 the sanity check `C == len(s2i)+1` (:44-45)
    fails for every pickle the writer produces; values beyond +-100 are CLIPPED to +-1 here (the
    other two loaders only count them, :47-65); 4th-root normalisation over (C,S1,S2,V) (:91-107)."""
    print("Start getting initial decompsoed parameters V C S1 S2...")
    dpath = os.path.join(data_dir, args.dataset)
    loader = load_glove_embed if args.embed_type == 'glove' else load_fasttext_embed
    pretrained_embed = loader(dpath, args.embed_dim)
    if args.random_embed:
        pretrained_embed = np.random.random(pretrained_embed.shape)

    automata_dicts = load_pkl(args.automata_path)
    print("Loading automata: {}".format(args.automata_path))
    per_seed = automata_dicts[args.seed]
    factor_dicts = per_seed[0][args.rank]
    factor_wildcard_dicts = per_seed[1][args.rank_wildcard]
    automata = automata_dicts['automata']

    V_embed, C_embed, S1, S2 = (factor_dicts[k] for k in ('V', 'C', 'S1', 'S2'))
    wildcard_tensor = factor_dicts['wildcard_tensor']
    wildcard_wildcard_tensor = factor_dicts['wildcard_wildcard_tensor']
    C_wildcard, S1_wildcard, S2_wildcard = (
        factor_wildcard_dicts[k] for k in ('C_wildcard', 'S1_wildcard', 'S2_wildcard'))

    if args.local_loss_func == 'CE1':            # sanity check (:44-45)
        assert C_embed.shape[0] == len(s2i) + 1

    print("Clipping corrupted values after decomposition")
    corrupt = 100
    mats = (V_embed, C_embed, S1, S2)
    print('Invalid Positive Values: {}'.format(int(sum(np.sum(m > corrupt) for m in mats))))
    for m in mats:
        m[m > corrupt] = 1
    print('Invalid Negative Values: {}'.format(int(sum(np.sum(m < -corrupt) for m in mats))))
    for m in mats:
        m[m < -corrupt] = -1

    n_state, rank = S1.shape
    final_vector = np.zeros(n_state)
    final_vector[automata['finalstates']] = 1
    start_vector = np.zeros(n_state)
    start_vector[automata['startstate']] = 1
    print("DFA states: {}".format(n_state))

    pretrain_embed_extend = np.append(pretrained_embed, np.zeros((1, args.embed_dim), dtype=np.float64), axis=0)
    V_embed_extend = np.append(V_embed, np.zeros((1, rank), dtype=np.float64), axis=0)
    priority_mat = create_mat_priority(s2i, args)

    if args.normalize_automata != 'none':       # (:89-119)
        print('Normalize automata decomposed parameters...')
        c_avg = get_average(C_embed, args.normalize_automata)
        s1_avg = get_average(S1, args.normalize_automata)
        s2_avg = get_average(S2, args.normalize_automata)
        v_avg = get_average(V_embed_extend, args.normalize_automata)
        factor = np.float_power(c_avg * s1_avg * s2_avg * v_avg, 1 / 4)
        S1 = S1 * (factor / s1_avg)
        S2 = S2 * (factor / s2_avg)
        C_embed = C_embed * (factor / c_avg)
        V_embed_extend = V_embed_extend * (factor / v_avg)
        if C_wildcard is not None:
            cw_avg = get_average(C_wildcard, args.normalize_automata)
            s1w_avg = get_average(S1_wildcard, args.normalize_automata)
            s2w_avg = get_average(S2_wildcard, args.normalize_automata)
            factor = np.float_power(cw_avg * s1w_avg * s2w_avg, 1 / 3)
            S1_wildcard = S1_wildcard * (factor / s1w_avg)
            S2_wildcard = S2_wildcard * (factor / s2w_avg)
            C_wildcard = C_wildcard * (factor / cw_avg)

    return (V_embed_extend, C_embed, S1, S2, pretrain_embed_extend, wildcard_tensor, wildcard_wildcard_tensor,
            final_vector, start_vector, priority_mat, C_wildcard, S1_wildcard, S2_wildcard)

"""Synthetic, schema-identical inputs for the tagging path.

No data ships with the reference (its ``data.zip`` is an external download), so every test,
fixture and benchmark here runs on generated inputs that follow the reference's on-disk
schemas:

  * ``dataset.pkl``        -- writer ``src_seq/data.py:412-418``
  * automaton dict         -- writer ``src_seq/wfa/create_dataset_automata.py:109-141``
  * ``IIID.automata.*.pkl`` -- writer ``src_seq/wfa/decompose_automata.py:373-431``
  * ``glove.<dim>.emb``     -- writer ``src_seq/data.py:80``

Everything is driven by a ``numpy.random.RandomState`` (frozen legacy stream) so fixtures
are reproducible from their seed.
"""
import numpy as np


# --------------------------------------------------------------------------- vocabularies
def make_vocab(n_words, n_numbers=6, with_punct=True):
    """t2i / i2t without the '<pad>' entry (the drivers append it last,
    reference train_onehot.py:47-48)."""
    words = []
    if with_punct:
        words += [',', '.', '?']
    nums = ['3', '17', '24', '25', '2021', '9.5', '0', '12', '100', '7'][:n_numbers]
    words += nums
    k = 0
    while len(words) < n_words:
        words.append('w{}'.format(k))
        k += 1
    words = words[:n_words]
    t2i = {w: i for i, w in enumerate(words)}
    i2t = {i: w for w, i in t2i.items()}
    return t2i, i2t


def make_slots(n_entity_types):
    """BIO slot vocabulary: 'o' plus b-/i- pairs (lower-cased like the reference's s2i)."""
    names = ['o']
    for k in range(n_entity_types):
        names.append('b-e{}'.format(k))
        names.append('i-e{}'.format(k))
    s2i = {s: i for i, s in enumerate(names)}
    i2s = {i: s for s, i in s2i.items()}
    return s2i, i2s


# --------------------------------------------------------------------------- rule automaton
def make_rule_automaton(t2i, s2i, n_states, rng, max_rule_len=4, words_per_edge=3,
                        special_edge_prob=0.15):
    """A slot-filling rule automaton in the reference's dict schema with the i-FST property
    (every state has a single incoming label, cf. ``wfa_convert.fix_inedge_node``).

    State 0 is start and final and carries a ``$<:>oo`` self loop (the "no rule fired" path);
    the last state is a shared accepting sink with the same loop; rules are chains
    ``0 -> q1 -> ... -> qk`` whose edges carry word sets and BIO labels.
    Returns (automaton_dict, rules) where rules lists the word/label chains (used to plant
    matches into generated sentences).
    """
    assert n_states >= 3
    words = [w for w in t2i if w not in ('<pad>',)]
    plain = [w for w in words if w.startswith('w')] or words
    ent = sorted({s[2:] for s in s2i if s.startswith('b-')})
    sink = n_states - 1
    trans = {0: {0: {'$<:>oo'}}, sink: {sink: {'$<:>oo'}}}
    finals = {0, sink}
    rules = []
    nxt = 1

    def add(fr, to, lab):
        trans.setdefault(fr, {}).setdefault(to, set()).add(lab)

    while nxt < sink:
        k = int(rng.randint(1, max_rule_len + 1))
        k = min(k, sink - nxt)
        e = ent[int(rng.randint(len(ent)))] if ent else None
        n_ctx = int(rng.randint(0, k)) if k > 1 else 0       # leading context edges tagged 'o'
        prev = 0
        chain = []
        for pos in range(k):
            q = nxt
            nxt += 1
            if pos < n_ctx or e is None:
                tag = 'o'
            elif pos == n_ctx:
                tag = 'b-' + e
            else:
                tag = 'i-' + e
            u = rng.rand()
            if u < special_edge_prob / 3:
                wset = ['%']
            elif u < 2 * special_edge_prob / 3:
                wset = ['&']
            elif u < special_edge_prob and pos > 0:
                wset = ['$']
            else:
                n_w = int(rng.randint(1, words_per_edge + 1))
                wset = [plain[int(i)] for i in rng.randint(0, len(plain), size=n_w)]
            for w in wset:
                add(prev, q, '{}<:>{}'.format(w, tag))
            chain.append((sorted(set(wset)), tag))
            prev = q
        # the chain end accepts, may repeat an i- label, and drains into the sink
        finals.add(prev)
        last_tag = chain[-1][1]
        if last_tag.startswith('i-') and rng.rand() < 0.5:
            for w in chain[-1][0]:
                add(prev, prev, '{}<:>{}'.format(w, last_tag))
        add(prev, sink, '$<:>oo')
        rules.append(chain)
    automaton = {
        'states': set(range(n_states)),
        'startstate': [0],
        'finalstates': sorted(finals),
        'transitions': trans,
    }
    return automaton, rules


def make_sentences(t2i, s2i, rules, n, max_len, rng, min_len=1, plant_prob=0.7,
                   zipf_a=1.1, full_length_rows=1):
    """Token-id sentences (lists of ints) + gold slot ids.  Tokens follow a Zipf law over the
    vocabulary; with probability `plant_prob` one rule's word chain is planted so that rules
    actually fire."""
    V = len(t2i)
    words = list(t2i.keys())
    ranks = np.arange(1, V + 1, dtype=np.float64)
    pz = ranks ** (-zipf_a)
    pz /= pz.sum()
    perm = rng.permutation(V)
    numbers = [w for w in words if w.replace('.', '', 1).isdigit()]
    puncts = [w for w in words if w in (',', '.', '?')]
    o_idx = s2i['o']
    queries, slots = [], []
    for k in range(n):
        ln = max_len if k < full_length_rows else int(rng.randint(min_len, max_len + 1))
        q = [int(perm[i]) for i in rng.choice(V, size=ln, p=pz)]
        s = [o_idx] * ln
        if rules and rng.rand() < plant_prob:
            chain = rules[int(rng.randint(len(rules)))]
            if len(chain) <= ln:
                at = int(rng.randint(0, ln - len(chain) + 1))
                for j, (wset, tag) in enumerate(chain):
                    w = wset[int(rng.randint(len(wset)))]
                    if w == '%':
                        w = numbers[int(rng.randint(len(numbers)))] if numbers else None
                    elif w == '&':
                        w = puncts[int(rng.randint(len(puncts)))] if puncts else None
                    elif w == '$':
                        w = None
                    if w is not None:
                        q[at + j] = t2i[w]
                    s[at + j] = s2i[tag]
        queries.append(q)
        slots.append(s)
    return queries, slots


def make_dataset(n_words, n_entity_types, n_states, seed, n_train=64, n_dev=32, n_test=32,
                 max_len=20):
    """A complete synthetic problem: vocabularies, automaton and a ``dataset.pkl``-shaped dict."""
    rng = np.random.RandomState(seed)
    t2i, i2t = make_vocab(n_words)
    s2i, i2s = make_slots(n_entity_types)
    automaton, rules = make_rule_automaton(t2i, s2i, n_states, rng)
    dset = {'t2i': t2i, 'i2t': i2t, 's2i': s2i, 'i2s': i2s}
    for name, n in (('train', n_train), ('dev', n_dev), ('test', n_test)):
        q, s = make_sentences(t2i, s2i, rules, n, max_len, rng)
        dset['query_' + name] = q
        dset['intent_' + name] = s          # slot ids despite the name (data.py:412-418)
    return dset, automaton, rules


# --------------------------------------------------------------------------- padded batches
def pad_batch(queries, max_len, pad_id):
    """reference utils.pad_dataset_1 (:28-56) for one batch -> (x[B,L] int64, lengths[B])."""
    keep = [q for q in queries if len(q) > 0]
    x = np.full((len(keep), max_len), pad_id, dtype=np.int64)
    lengths = np.zeros(len(keep), dtype=np.int64)
    for b, q in enumerate(keep):
        n = min(len(q), max_len)
        x[b, :n] = q[:n]
        lengths[b] = n
    return x, lengths


def random_batch(V, B, L, rng, min_len=5, zipf_a=1.1, full_length_rows=1):
    """Bench-style batch: Zipf tokens over the V-1 real ids, lengths ~ U[min_len, L] with
    `full_length_rows` rows of full length, pad id = V-1."""
    ranks = np.arange(1, V, dtype=np.float64)
    pz = ranks ** (-zipf_a)
    pz /= pz.sum()
    perm = rng.permutation(V - 1)
    lengths = rng.randint(min(min_len, L), L + 1, size=B).astype(np.int64)
    lengths[:full_length_rows] = L
    x = np.full((B, L), V - 1, dtype=np.int64)
    for b in range(B):
        n = int(lengths[b])
        x[b, :n] = perm[rng.choice(V - 1, size=n, p=pz)]
    return x, lengths


# --------------------------------------------------------------------------- direct tensors
def random_ifst_tensors(V, S, C, rng, edges_per_word=2.0, n_final=4, wildcard_moves=2):
    """Automaton-like dense i-FST tensors without going through an automaton dict
    (used at sizes where building an edge list would be slow): 0/1 ``T[V,S,S]`` with at most
    one successor per (word, from-state), wildcard self loops on the start state and the
    accepting states, one label per destination state, the ``oo`` column (C-1) on the
    wildcard states.  The last word id (V-1) is the pad row and stays all-zero.
    Path counts stay far below 2**24, so fp32 arithmetic on them is exact."""
    T = np.zeros((V, S, S), dtype=np.float32)
    n_edges = int(edges_per_word * (V - 1))
    w = rng.randint(0, V - 1, size=n_edges)
    s = rng.randint(0, S, size=n_edges)
    j = rng.randint(min(1, S - 1), S, size=n_edges)
    # keep at most one successor per (word, from-state): later draws are dropped
    _, first = np.unique(w.astype(np.int64) * S + s, return_index=True)
    T[w[first], s[first], j[first]] = 1.0
    W = np.zeros((S, S), dtype=np.float32)
    finals = np.unique(np.concatenate([[0, S - 1], rng.randint(min(1, S - 1), S, size=n_final)]))
    W[finals, finals] = 1.0
    for _ in range(wildcard_moves):
        a, b = int(rng.randint(min(1, S - 1), S)), int(finals[rng.randint(len(finals))])
        if a != b:
            W[a, b] = 1.0
    O = np.zeros((C, S), dtype=np.float32)
    lab = rng.randint(0, C - 1, size=S)
    O[lab, np.arange(S)] = 1.0
    O[:, finals] = 0.0
    O[C - 1, finals] = 1.0
    h0 = np.zeros(S, dtype=np.float32); h0[0] = 1.0
    hT = np.zeros(S, dtype=np.float32); hT[finals] = 1.0
    return T, W, O, h0, hT


def exact_cp_factors(T, rank=None, rng=None, noise=0.0):
    """An exact CP decomposition of a 0/1 language tensor: one rank-1 term per distinct
    (from-state, to-state) pair, ``T[w,s,j] = sum_r V[w,r] S1[s,r] S2[j,r]``.
    Optionally zero-pads to `rank` columns and adds gaussian noise.  Gives the decomposed
    model realistic, structured factors without tensorly (absent from this image)."""
    Vn, S, _ = T.shape
    pairs = np.argwhere(T.sum(0) > 0)
    R0 = len(pairs)
    R = R0 if rank is None else int(rank)
    assert R >= R0, "rank {} < number of (from,to) pairs {}".format(R, R0)
    Vf = np.zeros((Vn, R)); S1 = np.zeros((S, R)); S2 = np.zeros((S, R))
    for r, (s, j) in enumerate(pairs):
        Vf[:, r] = T[:, s, j]
        S1[s, r] = 1.0
        S2[j, r] = 1.0
    if noise and rng is not None:
        Vf += noise * rng.randn(*Vf.shape)
        S1 += noise * rng.randn(*S1.shape)
        S2 += noise * rng.randn(*S2.shape)
    return Vf, S1, S2


def make_iiid_pickle_dict(automaton, t2i, s2i, ranks, rng, noise=0.01, n_seeds=4,
                          dataset='MITR-BIO'):
    """A dict with the schema of ``IIID.automata.*.pkl`` (decompose_automata.py:373-431):
    {'automata': dict, seed: [ {rank: {'V','S1','S2','wildcard_mat'}},
                               {'output_mat'[C,S], 'output_wildcard_vector'[S]},
                               {'output_mat'[C+1,S], 'output_wildcard_vector'[S]} ]}.
    `t2i` must NOT contain '<pad>' (the loader appends the pad row, init_params.py:280-281)."""
    from .wfa.fsa_to_tensor import dfa_to_tensor_slot_single_wildcard
    lang, _, wild, out_mat, out_wild, _, _, _ = dfa_to_tensor_slot_single_wildcard(
        automaton, t2i, s2i, dataset=dataset)
    out = {'automata': automaton}
    for seed in range(n_seeds):
        per_rank = {}
        for R in ranks:
            Vf, S1, S2 = exact_cp_factors(lang, rank=R, rng=rng, noise=noise)
            per_rank[R] = {'V': Vf, 'S1': S1, 'S2': S2, 'wildcard_mat': wild.copy()}
        out[seed] = [per_rank,
                     {'output_mat': out_mat[:-1].copy(), 'output_wildcard_vector': out_wild.copy()},
                     {'output_mat': out_mat.copy(), 'output_wildcard_vector': out_wild.copy()}]
    return out


def exact_cp_factors_4d(T4, rank=None, rng=None, noise=0.0):
    """Exact CP factors of a 0/1 4th-order language tensor, one rank-1 term per labelled edge:
    ``T4[w,c,s,j] = sum_r V[w,r] C[c,r] S1[s,r] S2[j,r]``."""
    Vn, C, S, _ = T4.shape
    trip = np.argwhere(T4.sum(0) > 0)
    R = len(trip) if rank is None else int(rank)
    assert R >= len(trip), "rank {} < number of labelled edges {}".format(R, len(trip))
    Vf = np.zeros((Vn, R)); Cf = np.zeros((C, R)); S1 = np.zeros((S, R)); S2 = np.zeros((S, R))
    for r, (c, s, j) in enumerate(trip):
        Vf[:, r] = T4[:, c, s, j]
        Cf[c, r] = 1.0; S1[s, r] = 1.0; S2[j, r] = 1.0
    if noise and rng is not None:
        for a in (Vf, Cf, S1, S2):
            a += noise * rng.randn(*a.shape)
    return Vf, Cf, S1, S2


def make_d_pickle_dict(automaton, t2i, s2i, ranks, wildcard_ranks, rng, noise=0.01, n_seeds=4,
                       dataset='MITR-BIO'):
    """A dict with the schema of ``D.automata.*.pkl`` written by decompose_automata
    (decompose_automata.py:30-146):
    {'automata': dict, seed: [ {rank: {'V','C','S1','S2','wildcard_tensor','wildcard_wildcard_tensor'}},
                               {rank_w: {'C_wildcard'[C,RW], 'S1_wildcard','S2_wildcard'[S,RW]}} ]}."""
    from .wfa.fsa_to_tensor import dfa_to_tensor_slot_new_wildcard
    T4, _, W4, WW, _, _, _ = dfa_to_tensor_slot_new_wildcard(automaton, t2i, s2i, dataset=dataset)
    out = {'automata': automaton}
    for seed in range(n_seeds):
        per_rank, per_rw = {}, {}
        for R in ranks:
            Vf, Cf, S1, S2 = exact_cp_factors_4d(T4, rank=R, rng=rng, noise=noise)
            per_rank[R] = {'V': Vf, 'C': Cf, 'S1': S1, 'S2': S2, 'wildcard_tensor': W4.copy(),
                           'wildcard_wildcard_tensor': WW.copy()}
        for RW in wildcard_ranks:
            Cw, S1w, S2w = exact_cp_factors(W4, rank=RW, rng=rng, noise=noise)
            per_rw[RW] = {'C_wildcard': Cw, 'S1_wildcard': S1w, 'S2_wildcard': S2w}
        out[seed] = [per_rank, per_rw]
    return out


def make_iid_pickle_dict(automaton, t2i, s2i, ranks, output_ranks, rng, noise=0.01, n_seeds=4,
                         dataset='MITR-BIO'):
    """A dict with the schema of ``IID.automata.*.pkl`` written by decompose_automata_independent
    (decompose_automata.py:148-300):
    {'automata': dict, seed: [ {rank: {'V','S1','S2','wildcard_mat'}},
                               {rank_o: {'C_output'[C,RO], 'S1_output','S2_output'[S,RO], 'wildcard_output'[S,S]}},
                               {rank_o: {'C_output'[C+1,RO], ..., 'wildcard_output': None}} ]}   # CE1."""
    from .wfa.fsa_to_tensor import dfa_to_tensor_slot_independent_wildcard
    lang, _, wild, out_ten, _, _, _, _ = dfa_to_tensor_slot_independent_wildcard(
        automaton, t2i, s2i, dataset=dataset)
    out = {'automata': automaton}
    for seed in range(n_seeds):
        per_rank, per_ro, per_ro_w = {}, {}, {}
        for R in ranks:
            Vf, S1, S2 = exact_cp_factors(lang, rank=R, rng=rng, noise=noise)
            per_rank[R] = {'V': Vf, 'S1': S1, 'S2': S2, 'wildcard_mat': wild.copy()}
        for RO in output_ranks:
            Cf, S1o, S2o = exact_cp_factors(out_ten[:-1], rank=RO, rng=rng, noise=noise)
            per_ro[RO] = {'C_output': Cf, 'S1_output': S1o, 'S2_output': S2o,
                          'wildcard_output': out_ten[-1].copy()}
            Cf, S1o, S2o = exact_cp_factors(out_ten, rank=RO, rng=rng, noise=noise)
            per_ro_w[RO] = {'C_output': Cf, 'S1_output': S1o, 'S2_output': S2o, 'wildcard_output': None}
        out[seed] = [per_rank, per_ro, per_ro_w]
    return out


def random_decomposed_params(V, S, C, R, D, rng, scale=None, contractive=False):
    """Dense gaussian factors for size/throughput runs of the decomposed path.

    contractive=True: a WELL-CONDITIONED model for parity checks at full sequence length.  With the default
    scales the per-token transition matrix (sum_r v_r S1 S2^T + W) * o has spectral radius above one and
    the 64-step tanh recurrence is chaotic: two float32 evaluations that differ only in summation order
    (or float32 vs float64) drift apart by ~1e-2, so no implementation -- the reference included -- can be
    held to 1e-4 on it.  The contractive variant follows the structure of real decomposed i-FSTs more
    closely: every state carries exactly one label (output_mat has one 1 per destination state,
    fsa_to_tensor.py:586, so o = sum_c C[c,:] = 1), the factor part has spectral radius ~0.45 and the
    wildcard part ~0.4, except for
    the self loops of the start state and of a final sink (weight 1.5: tanh fixed point 0.86 with slope 0.4), which keep
    the states -- and so the scores -- of order one: rounding differences decay instead of growing."""
    if scale is None:
        scale = 0.9 / float(np.cbrt(S * R) ** 0.5)
        if contractive:
            scale = float((0.45 / np.sqrt(S * R)) ** (1.0 / 3.0))
    p = {
        'V_embed': (rng.randn(V, R) * scale).astype(np.float64),
        'S1': (rng.randn(S, R) * scale).astype(np.float64),
        'S2': (rng.randn(S, R) * scale).astype(np.float64),
        'wildcard_mat': (rng.rand(S, S) < 2.0 / S).astype(np.float64) * (0.2 if contractive else 0.5),
        'C_output_mat': (rng.rand(C, S) < 1.5 / C).astype(np.float64),
        'wildcard_output_vector': np.zeros(S),
        'embed': (rng.randn(V, D) * 0.3).astype(np.float64),
    }
    if contractive:
        Cm = np.zeros((C, S))
        Cm[rng.randint(0, C, size=S), np.arange(S)] = 1.0
        p['C_output_mat'] = Cm
        # the start state and a final sink keep a wildcard self loop (the "no rule fired" path of a real automaton);
        # 1.5 puts their tanh fixed point at 0.86 with slope 0.4: the states stay alive AND rounding noise decays
        p['wildcard_mat'][0, 0] = 1.5
        p['wildcard_mat'][S - 1, S - 1] = 1.5
    p['V_embed'][V - 1] = 0.0
    p['embed'][V - 1] = 0.0
    h0 = np.zeros(S); h0[0] = 1.0
    hT = (rng.rand(S) < 0.1).astype(np.float64); hT[0] = 1.0
    if contractive:
        hT[S - 1] = 1.0
    p['start_vector'] = h0
    p['final_vector'] = hT
    return p


def snips_sized_model(R, farnn, crf, seed=1234, S=104, V=11000, C=73):
    """BASELINE configs[2] as `bench.py --workload decomp` builds it (same seed, same generator), plus the GRU-style
    gates (model_decompose_single.py:93-123) and the CRF rows / transitions (:78-79, crf.py:31-46) on demand.  One source
    for bench.py's shape, tests/test_gpu_parity_bench_size.py and tests/golden/make_golden_bench.py (which feeds
    exactly these arrays to the reference's FARNN_S_D_W_I_S).  Returns (V, q, gates, crf_transitions): `q` is the
    parameter dict the tests' checker takes for the decomposed i-FST (nl = 2: tanh, semiring = 0: sum)."""
    wrng = np.random.RandomState(seed)
    p = random_decomposed_params(V, S, C, R, 100, wrng, contractive=True)
    f = lambda a: np.asarray(a, np.float32)                       # noqa: E731
    Cout = f(p['C_output_mat'])
    tr = None
    if crf:            # two extra rows for START / STOP, small random values; default transitions + noise
        Cout = np.concatenate([Cout, (wrng.rand(2, S) * 0.01).astype(np.float32)], 0)
        K = C + 2
        tr0 = np.zeros((K, K), np.float32)                        # CRF.__init__ (crf.py:31-46)
        tr0[:, K - 2] = -10000.0
        tr0[K - 1, :] = -10000.0
        tr = tr0 + (wrng.randn(K, K) * 1.0).astype(np.float32)
    q = {'Vgen': f(p['V_embed']), 'S1': f(p['S1']), 'S2': f(p['S2']), 'W': f(p['wildcard_mat']), 'Cout': Cout,
         'h0': f(p['start_vector']), 'hT': f(p['final_vector']), 'farnn': farnn, 'nl': 2, 'semiring': 0, 'sig_k': 5}
    gates = None
    if farnn:
        gates = {'Wss1': f(wrng.randn(S, S) * 0.03), 'Wrs1': f(wrng.randn(R, S) * 0.03), 'bs1': f(np.full(S, 1.0))}
        if farnn == 2:
            gates.update(Wss2=f(wrng.randn(S, S) * 0.03), Wrs2=f(wrng.randn(R, S) * 0.03), bs2=f(np.full(S, 1.0)))
        q.update(gates)
    return V, q, gates, tr


def planted_rule_ifst(seed=1234, V=11000, S=104, C=73, max_pairs=250, max_rule_len=5, words_per_pair=(5, 60)):
    """A rule automaton at SNIPS-BIO size in edge-list form, with an EXACT rank-`max_pairs` CP form (the layout of
    decompose_automata.py:373-431: one rank-1 term per (from-state, to-state) pair -- V_embed[:, r] = 1[word in the pair's
    set], S1[:, r] = 1[from], S2[:, r] = 1[to], wildcard_mat = W): the decomposed i-FST built from these factors computes,
    with update_nonlinear = none, exactly the path counts of the onehot i-FST of the same automaton.

    State 0: start, final, wildcard self loop (`oo`).  State S-1: accepting sink, wildcard self loop (`oo`).  States 1..S-2:
    rule chains 0 -> q1 -> ... -> qk (k = 1..max_rule_len) whose states carry one BIO label each (the i-FST property: the label
    sits on the destination state, fsa_to_tensor.py:586); qk is final and moves to the sink on any token (a wildcard edge), and
    the sink re-enters every chain on the chain's first word set, so a sentence can match several rules; skip edges
    q_i -> q_{i+2} fill the pair budget.  The outgoing pairs of one state have disjoint word sets.

    Returns a dict: edge list (word, frm, to: int32; word = -1 for wildcard edges), `state_label` [S] (the label column of every
    state; C-1 = `oo`), h0, hT, W [S,S], O [C,S], `pairs` [(from, to)], `pair_words` (list of int arrays), `chains` (list of
    lists of (pair index, state)), and the exact factors Vgen [V,R], S1 [S,R], S2 [S,R] with R = max_pairs."""
    rng = np.random.RandomState(seed)
    sink = S - 1
    n_ent = (C - 2) // 2                          # labels: 0 = 'o', 1 + 2e = b-e, 2 + 2e = i-e; column C-1 = `oo`
    state_label = np.full(S, C - 1, np.int64)
    chains, pairs, pair_of = [], [], {}
    nxt = 1

    def pair(a, b):
        if (a, b) not in pair_of:
            pair_of[(a, b)] = len(pairs)
            pairs.append((a, b))
        return pair_of[(a, b)]

    while nxt < sink:
        k = min(int(rng.randint(1, max_rule_len + 1)), sink - nxt)
        e = int(rng.randint(n_ent))
        n_ctx = int(rng.randint(0, 2)) if k > 1 else 0           # a leading context state tagged 'o'
        chain, prev = [], 0
        for pos in range(k):
            q = nxt; nxt += 1
            state_label[q] = 0 if pos < n_ctx else (1 + 2 * e if pos == n_ctx else 2 + 2 * e)
            chain.append((pair(prev, q), q))
            prev = q
        chains.append(chain)
    first_pairs = [c[0][0] for c in chains]
    reentry = {c[0][1]: pair(sink, c[0][1]) for c in chains}     # sink -> q1: the same word set as 0 -> q1
    skips = [(c[i][1], c[i + 2][1]) for c in chains for i in range(len(c) - 2)]
    for idx in rng.permutation(len(skips)):
        if len(pairs) >= max_pairs:
            break
        pair(*skips[idx])
    assert len(pairs) <= max_pairs, (len(pairs), max_pairs)
    # word sets: disjoint among the outgoing pairs of a state; a chain's first set is shared by its entry from 0 and from the sink
    pair_words = [None] * len(pairs)
    by_from = {}
    for r, (a, b) in enumerate(pairs):
        by_from.setdefault(a, []).append(r)
    for a, rs in by_from.items():
        if a == sink:
            continue
        sizes = rng.randint(words_per_pair[0], words_per_pair[1] + 1, size=len(rs))
        pool = rng.permutation(V - 1)[:int(sizes.sum())]
        at = 0
        for r, n in zip(rs, sizes):
            pair_words[r] = np.sort(pool[at:at + n]).astype(np.int64)
            at += n
    for c in chains:
        pair_words[reentry[c[0][1]]] = pair_words[c[0][0]]
    word = np.concatenate([pair_words[r] for r in range(len(pairs))])
    frm = np.concatenate([np.full(len(pair_words[r]), pairs[r][0]) for r in range(len(pairs))])
    to = np.concatenate([np.full(len(pair_words[r]), pairs[r][1]) for r in range(len(pairs))])
    W = np.zeros((S, S), np.float32)
    W[0, 0] = 1.0; W[sink, sink] = 1.0
    hT = np.zeros(S, np.float32); hT[0] = 1.0; hT[sink] = 1.0
    for c in chains:
        W[c[-1][1], sink] = 1.0
        hT[c[-1][1]] = 1.0
    h0 = np.zeros(S, np.float32); h0[0] = 1.0
    O = np.zeros((C, S), np.float32)
    O[state_label, np.arange(S)] = 1.0
    R = int(max_pairs)
    Vgen = np.zeros((V, R), np.float32); S1 = np.zeros((S, R), np.float32); S2 = np.zeros((S, R), np.float32)
    for r, (a, b) in enumerate(pairs):
        Vgen[pair_words[r], r] = 1.0
        S1[a, r] = 1.0
        S2[b, r] = 1.0
    wa, wb = np.nonzero(W)
    return {'V': V, 'S': S, 'C': C, 'word': np.concatenate([word, np.full(len(wa), -1)]).astype(np.int32),
            'frm': np.concatenate([frm, wa]).astype(np.int32), 'to': np.concatenate([to, wb]).astype(np.int32),
            'state_label': state_label, 'h0': h0, 'hT': hT, 'W': W, 'O': O, 'pairs': pairs, 'pair_words': pair_words,
            'chains': chains, 'first_pairs': first_pairs, 'Vgen': Vgen, 'S1': S1, 'S2': S2}


def planted_rule_batch(A, B, L, seed, min_len=5, fill=0.8, max_rules=9):
    """A bench-shaped batch (lengths U[min_len, L], one full-length row, Zipf filler tokens, pad id V-1) in which rules of the
    automaton `A` (planted_rule_ifst) really fire: in about `fill` of the sequences chains are planted back to back (one
    filler token between two of them: the wildcard move into the sink), at most `max_rules` per sequence (the number of
    accepting paths doubles with every planted rule; nine keep every count far below 2**24)."""
    rng = np.random.RandomState(seed)
    V = A['V']
    x, lengths = random_batch(V, B, L, rng, min_len=min_len)
    for b in range(B):
        if rng.rand() >= fill:
            continue
        n = int(lengths[b])
        t = int(rng.randint(0, 4))
        for _ in range(max_rules):
            c = A['chains'][int(rng.randint(len(A['chains'])))]
            if t + len(c) > n:
                break
            for r, _q in c:
                ws = A['pair_words'][r]
                x[b, t] = ws[int(rng.randint(len(ws)))]
                t += 1
            t += 1 + int(rng.randint(0, 2))            # the token(s) the sink consumes before the next rule
    return x, lengths


def exact_case_transitions(C, rng, scale=0.3):
    """CRF transitions over C labels + START / STOP for the automaton-derived decomposed model: CRF.__init__'s defaults
    (crf.py:31-46) plus noise small enough that the emission scores still decide most positions"""
    K = C + 2
    tr = np.zeros((K, K), np.float32)
    tr[:, K - 2] = -10000.0
    tr[K - 1, :] = -10000.0
    return tr + (rng.randn(K, K) * scale).astype(np.float32)


def exact_case_gates(S, R, farnn, rng):
    """GRU-style gates as the reference initialises them around an automaton's factors (model_decompose_single.py:93-123: small
    weights, a positive bias -- the gates start nearly open)"""
    f = lambda a: np.asarray(a, np.float32)                       # noqa: E731
    g = {'Wss1': f(rng.randn(S, S) * 0.03), 'Wrs1': f(rng.randn(R, S) * 0.03), 'bs1': f(np.full(S, 1.0))}
    if farnn == 2:
        g.update(Wss2=f(rng.randn(S, S) * 0.03), Wrs2=f(rng.randn(R, S) * 0.03), bs2=f(np.full(S, 1.0)))
    return g


def dense_from_edges(A):
    """the dense [V,S,S] language tensor of a planted_rule_ifst automaton (476 MB of float32 at SNIPS size: generator and
    oracle side only)"""
    T = np.zeros((A['V'], A['S'], A['S']), np.float32)
    m = A['word'] >= 0
    T[A['word'][m], A['frm'][m], A['to'][m]] = 1.0
    return T


def atis_sized_crf_model(seed=1234, V=950, S=71, C=128, tr_scale=0.1):
    """BASELINE configs[3] as `bench.py --workload ifst_crf` builds it: the onehot i-FST and the transitions of a CRF over
    its C labels + START / STOP (`tr_scale`: the spread of the transition scores; the bench-size fixture uses a larger one so
    that the decoded path really differs from the per-position arg-max)."""
    wrng = np.random.RandomState(seed)
    T, W, O, h0, hT = random_ifst_tensors(V, S, C, wrng)
    K = C + 2
    tr = (wrng.randn(K, K) * tr_scale).astype(np.float32)
    tr[:, K - 2] = -10000.0
    tr[K - 1, :] = -10000.0
    return T, W, O, h0, hT, tr


# --------------------------------------------------------------------------- on-disk trees
def write_dataset_tree(root, dataset='ATIS-BIO', n_words=60, n_entity_types=4, n_states=20, seed=0,
                       n_train=48, n_dev=24, n_test=24, max_len=16, embed_dim=16, ranks=(100,),
                       output_ranks=(70,)):
    """Write a complete, schema-identical data directory for the CLI drivers:
        <root>/<dataset>/dataset.pkl
        <root>/<dataset>/glove.<dim>.emb
        <root>/<dataset>/automata/synthetic.ID{0,1,2}           (automaton dict, one per --independent)
        <root>/<dataset>/automata/IIID.automata.synthetic.pkl   (decomposed i-FST)
        <root>/<dataset>/automata/IID.automata.synthetic.pkl    (decomposed independent=1)
        <root>/<dataset>/automata/D.automata.synthetic.pkl      (decomposed independent=0)
    Returns a dict of the paths and the generated objects."""
    import os
    import pickle
    rng = np.random.RandomState(seed + 1000)
    dset, automaton, rules = make_dataset(n_words, n_entity_types, n_states, seed,
                                          n_train=n_train, n_dev=n_dev, n_test=n_test, max_len=max_len)
    ddir = os.path.join(root, dataset)
    adir = os.path.join(ddir, 'automata')
    os.makedirs(adir, exist_ok=True)
    with open(os.path.join(ddir, 'dataset.pkl'), 'wb') as f:
        pickle.dump(dset, f)
    with open(os.path.join(ddir, 'glove.{}.emb'.format(embed_dim)), 'wb') as f:
        pickle.dump(rng.randn(len(dset['t2i']), embed_dim) * 0.5, f)
    paths = {'data_dir': root + ('' if root.endswith('/') else '/'), 'dataset_dir': ddir}
    for ind in (0, 1, 2):
        p = os.path.join(adir, 'synthetic.ID{}'.format(ind))
        with open(p, 'wb') as f:
            pickle.dump({'automata': automaton} if ind == 2 else automaton, f)   # both wrappers occur
        paths['ID{}'.format(ind)] = p
    iiid = make_iiid_pickle_dict(automaton, dset['t2i'], dset['s2i'], ranks=list(ranks), rng=rng)
    p = os.path.join(adir, 'IIID.automata.synthetic.pkl')
    with open(p, 'wb') as f:
        pickle.dump(iiid, f)
    paths['IIID'] = p
    iid = make_iid_pickle_dict(automaton, dset['t2i'], dset['s2i'], ranks=list(ranks),
                               output_ranks=list(output_ranks), rng=rng)
    p = os.path.join(adir, 'IID.automata.synthetic.pkl')
    with open(p, 'wb') as f:
        pickle.dump(iid, f)
    paths['IID'] = p
    dd = make_d_pickle_dict(automaton, dset['t2i'], dset['s2i'], ranks=list(ranks),
                            wildcard_ranks=list(output_ranks), rng=rng)
    p = os.path.join(adir, 'D.automata.synthetic.pkl')
    with open(p, 'wb') as f:
        pickle.dump(dd, f)
    paths['D'] = p
    return {'paths': paths, 'dset': dset, 'automaton': automaton, 'rules': rules}

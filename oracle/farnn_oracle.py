"""CPU ORACLE for the FA-RNN forward tagging path -- TEST INFRASTRUCTURE ONLY.

This file is a plain numpy (float32) restatement of the reference's algorithm for the
forward tagging path of jeffchy/RE2NN-SEQ.  It is NOT part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``re2nn-seq_amd``) never imports it and
fails loudly when the HIP library is missing.

Parity pin: every function below is checked against outputs captured from the imported
reference classes (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) by
``tests/test_oracle_golden.py``.  The reference ships no golden vectors of its own
(SURVEY.md section 4), so the captured fixtures are the pin.

Each function cites the reference file:line it follows (paths relative to the reference
checkout, ``src_seq/...``).

Conventions (identical to the reference):
  x        int64 [B, L]   token ids, padded with the pad id (V-1)
  lengths  int64 [B]      1..L
  scores   float32 [B, L, C]
The reference iterates the recurrence over ALL L positions (pads included) for the onehot
models (model_onehot.py:372) and over ``lengths.max()`` positions for the decomposed
models (model_decompose_single.py:221); this restatement does the same so that
``forward_RE`` (pads kept) can be checked too.
"""
import contextlib

import numpy as np

F32 = np.float32


@contextlib.contextmanager
def precision(dtype):
    """Evaluate the restatement in another float type (tests only).  float64 gives the value that float32 implementations
    -- the reference's, numpy's at another batch size, the HIP kernels' -- scatter around: on a locally sensitive sequence
    two float32 evaluations can sit further from each other than either sits from it."""
    global F32
    old, F32 = F32, dtype
    try:
        yield
    finally:
        F32 = old

NL_NONE, NL_RELU, NL_TANH, NL_RELUTANH, NL_SIGMOID = 0, 1, 2, 3, 4
NL_CODES = {'none': NL_NONE, 'relu': NL_RELU, 'tanh': NL_TANH, 'relutanh': NL_RELUTANH,
            'sigmoid': NL_SIGMOID}
SEMIRING_SUM, SEMIRING_MAX = 0, 1


def _nl(v, nl):
    """update_nonlinear dispatch, model_onehot.py:379-386 / model_decompose_single.py:182-189."""
    if nl == NL_RELU:
        return np.maximum(v, F32(0))
    if nl == NL_TANH:
        return np.tanh(v).astype(F32)
    if nl == NL_RELUTANH:
        return np.tanh(np.maximum(v, F32(0))).astype(F32)
    if nl == NL_SIGMOID:
        return (F32(1) / (F32(1) + np.exp(-v))).astype(F32)
    return v


def semiring_vm(h, Tr, semiring):
    """h[B,S] (x) Tr[B,S,S] -> [B,S]; utils.py:192-199 (_matmul / _maxmul)."""
    if semiring == SEMIRING_MAX:
        return (h[:, :, None] * Tr).max(axis=1).astype(F32)
    return np.einsum('bs,bsj->bj', h, Tr).astype(F32)


def reverse_prefix(a, lengths):
    """utils.py:183-189 `reverse`: flip the first lengths[b] entries of every row."""
    out = a.copy()
    for b in range(a.shape[0]):
        n = int(lengths[b])
        out[b, :n] = a[b, :n][::-1]
    return out


def flatten(a, lengths):
    """utils.py:153-164 `flatten`: concatenate the valid prefix of every row."""
    return np.concatenate([a[b, :int(lengths[b])] for b in range(a.shape[0])], axis=0)


def priority(scores, P):
    """priority.py:20-30: scores @ P (+ zero bias)."""
    return (scores.astype(F32) @ P.astype(F32)).astype(F32)


def expand_priority(C, priority_mat):
    """priority.py:6-18: identity[C,C] with the given matrix in the top-left corner."""
    base = np.eye(C, dtype=F32)
    if priority_mat is not None:
        pm = np.asarray(priority_mat, dtype=F32)
        n = pm.shape[0]
        base[:n, :n] = pm
    return base


# --------------------------------------------------------------------------- onehot i-FST
def onehot_ifst_scores(T, W, O, h0, hT, x, lengths, nl=NL_NONE, semiring=SEMIRING_SUM, P=None):
    """FARNN_S_O_I_S.forward_score, model_onehot.py:351-428 (independent=2, CE1).

    T [V,S,S] language tensor, W [S,S] wildcard matrix, O [C,S] output matrix.
    Returns scores [B,L,C] for all L positions (pads included, like the reference).
    """
    T = np.asarray(T, F32); W = np.asarray(W, F32); O = np.asarray(O, F32)
    h0 = np.asarray(h0, F32); hT = np.asarray(hT, F32)
    B, L = x.shape
    S = h0.shape[0]
    Tf = T + W                                   # :366
    o = O.sum(0).astype(F32)                     # :368 (CE1)
    xb = reverse_prefix(x, lengths)              # :359
    hf = np.repeat(h0[None], B, 0)
    hb = np.repeat(hT[None], B, 0)
    fw = np.zeros((B, L + 1, S), F32); fw[:, 0] = h0
    bw = np.zeros((B, L + 1, S), F32); bw[:, 0] = hT
    for i in range(L):
        hf = semiring_vm(hf, Tf[x[:, i]], semiring)       # :375-377
        hf = _nl(hf * o, nl)                               # :378-386
        fw[:, i + 1] = hf
        hb = hb * o                                        # :393
        hb = semiring_vm(hb, Tf[xb[:, i]].transpose(0, 2, 1), semiring)   # :394
        hb = _nl(hb, nl)
        bw[:, i + 1] = hb
    rb = reverse_prefix(bw, np.asarray(lengths) + 1)       # :415
    scores = np.zeros((B, L, O.shape[0]), F32)
    for i in range(L):
        ab = fw[:, i + 1] * rb[:, i + 1]                   # :419-420, :347
        sc = np.einsum('cs,bs->bc', O, ab).astype(F32)     # :348
        if P is not None:
            sc = priority(sc, P)                           # :423-424
        scores[:, i] = sc
    return scores


# --------------------------------------------------------------------------- onehot FST (4-D)
def onehot_fst4_scores(T4, W4, h0, hT, x, lengths, semiring=SEMIRING_SUM, P=None):
    """FARNN_S_O.forward_score, model_onehot.py:66-129 (independent=0, CE1).

    T4 [V,C,S,S], W4 [C,S,S].  relu is applied unconditionally (:93-94, :100-101).
    """
    T4 = np.asarray(T4, F32); W4 = np.asarray(W4, F32)
    h0 = np.asarray(h0, F32); hT = np.asarray(hT, F32)
    B, L = x.shape
    C, S = W4.shape[0], W4.shape[1]
    Ts = T4.sum(1) + W4.sum(0)                   # :82
    A = T4 + W4                                  # :87
    xb = reverse_prefix(x, lengths)
    hf = np.repeat(h0[None], B, 0); hb = np.repeat(hT[None], B, 0)
    fw = np.zeros((B, L + 1, S), F32); fw[:, 0] = h0
    bw = np.zeros((B, L + 1, S), F32); bw[:, 0] = hT
    for i in range(L):
        hf = np.maximum(semiring_vm(hf, Ts[x[:, i]], semiring), F32(0))
        fw[:, i + 1] = hf
        hb = np.maximum(semiring_vm(hb, Ts[xb[:, i]].transpose(0, 2, 1), semiring), F32(0))
        bw[:, i + 1] = hb
    rb = reverse_prefix(bw, np.asarray(lengths) + 1)
    scores = np.zeros((B, L, C), F32)
    for i in range(L):
        Tr = A[x[:, i]]                                       # :116  B,C,S,S
        alpha = fw[:, i]                                      # :117  state BEFORE token i
        beta = rb[:, i + 1]                                   # :118
        tmp = Tr * alpha[:, None, :, None]                    # :119
        tmp = tmp * beta[:, None, None, :]                    # :120
        sc = np.maximum(tmp, F32(0)).sum(axis=(2, 3)).astype(F32)   # :121-122
        if P is not None:
            sc = priority(sc, P)
        scores[:, i] = sc
    return scores


# --------------------------------------------------------------------------- onehot independent=1
def onehot_ind1_scores(T, W, Oten, h0, hT, x, lengths, semiring=SEMIRING_SUM, P=None,
                       mask_by_output=False):
    """FARNN_S_O_I.forward_score, model_onehot.py:235-306.

    T [V,S,S], W [S,S], Oten [C,S,S].  relu always.  `mask_by_output` is the
    `args.independent == 2` branch (:259-262, :271-274).
    """
    T = np.asarray(T, F32); W = np.asarray(W, F32); Oten = np.asarray(Oten, F32)
    h0 = np.asarray(h0, F32); hT = np.asarray(hT, F32)
    B, L = x.shape
    C, S = Oten.shape[0], Oten.shape[1]
    Tf = T + W
    osum = Oten.sum(0)
    xb = reverse_prefix(x, lengths)
    hf = np.repeat(h0[None], B, 0); hb = np.repeat(hT[None], B, 0)
    fw = np.zeros((B, L + 1, S), F32); fw[:, 0] = h0
    bw = np.zeros((B, L + 1, S), F32); bw[:, 0] = hT
    for i in range(L):
        Tr = Tf[x[:, i]]
        Trb = Tf[xb[:, i]]
        if mask_by_output:
            Tr = Tr * osum
            Trb = Trb * osum
        hf = np.maximum(semiring_vm(hf, Tr, semiring), F32(0))
        fw[:, i + 1] = hf
        hb = np.maximum(semiring_vm(hb, Trb.transpose(0, 2, 1), semiring), F32(0))
        bw[:, i + 1] = hb
    rb = reverse_prefix(bw, np.asarray(lengths) + 1)
    scores = np.zeros((B, L, C), F32)
    for i in range(L):
        Tr = Tf[x[:, i]]
        ab = fw[:, i][:, :, None] * rb[:, i + 1][:, None, :]      # :230
        abt = ab * Tr                                               # :231
        sc = np.einsum('csj,bsj->bc', Oten, abt).astype(F32)        # :232
        if P is not None:
            sc = priority(sc, P)
        scores[:, i] = sc
    return scores


# --------------------------------------------------------------------------- decode
def decode_argmax(scores, threshold, o_idx):
    """local_decode / forward_RE / decode(non-CRF), CE1:
    model_onehot.py:148-180, model_decompose.py:363-369.
    Clamp the last column to `threshold`, first-index argmax, map K-1 -> o_idx."""
    s = np.array(scores, dtype=F32, copy=True)
    K = s.shape[-1]
    s[..., K - 1] = np.minimum(s[..., K - 1], F32(threshold))
    pred = s.argmax(axis=-1).astype(np.int64)          # numpy argmax = first maximal index
    pred[pred == K - 1] = o_idx
    return pred


def crf_default_transitions(tagset):
    """CRF.__init__, crf.py:31-46."""
    K = tagset + 2
    tr = np.zeros((K, K), F32)
    tr[:, K - 2] = -10000.0
    tr[K - 1, :] = -10000.0
    return tr


def viterbi_paths(feats, lengths, tr):
    """CRF._viterbi_decode, crf.py:102-195, restated per sequence over valid positions.

    feats [B,L,K] (K = tagset+2), tr [K,K].  Returns int64 [B,L] with zeros at pads
    (the reference leaves junk at pads; only valid positions are compared).
    Association kept as in the reference: (feat[j] + tr[i,j]) + part[i]  (crf.py:123,145).
    """
    feats = np.asarray(feats, F32); tr = np.asarray(tr, F32)
    B, L, K = feats.shape
    START, STOP = K - 2, K - 1
    out = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(lengths[b])
        part = (feats[b, 0] + tr[START]).astype(F32)         # :135
        bps = []
        for t in range(1, n):
            cur = (feats[b, t][None, :] + tr) + part[:, None]    # :145
            bps.append(cur.argmax(axis=0))                        # :149 first max over i
            part = cur.max(axis=0).astype(F32)
        last = part[:, None] + tr                                 # :168
        ptr = int(last[:, STOP].argmax())                         # :169,:177
        out[b, n - 1] = ptr
        for t in range(n - 2, -1, -1):
            ptr = int(bps[t][ptr])
            out[b, t] = ptr
    return out


def decode_crf(scores, lengths, tr, threshold, o_idx):
    """FARNN_S_D_W.decode, CRF branch, CE1: model_decompose.py:349-356.
    Clamp column K-3, Viterbi, map K-3 -> o_idx.  Returns [B,L] (valid positions only)."""
    s = np.array(scores, dtype=F32, copy=True)
    K = s.shape[-1]
    s[..., K - 3] = np.minimum(s[..., K - 3], F32(threshold))
    p = viterbi_paths(s, lengths, tr)
    p[p == K - 3] = o_idx
    return p


# --------------------------------------------------------------------------- decomposed i-FST
def generalized_vocab_table(V_embed, E, G, beta_vec, add_nl=NL_NONE):
    """get_generalized_v_embed_vec for every word at once, model_decompose.py:222-241.
    Vgen[w] = V_embed[w]*beta + nl_add(E[w] @ G)*(1-beta)."""
    V_embed = np.asarray(V_embed, F32); E = np.asarray(E, F32); G = np.asarray(G, F32)
    beta_vec = np.asarray(beta_vec, F32)
    gen = _nl((E @ G).astype(F32), add_nl)
    return (V_embed * beta_vec + gen * (F32(1) - beta_vec)).astype(F32)


def _sig(v, k):
    with np.errstate(over='ignore'):        # exp(+big) -> inf -> 1/(1+inf) = 0, as torch.sigmoid saturates
        return (F32(1) / (F32(1) + np.exp(-(v * F32(k))))).astype(F32)


def decomp_ifst_step(h, v, h_init, o, p, fwd):
    """FARNN_S_D_W_I_S.get_forward_score, model_decompose_single.py:138-200.
    p: dict with S1,S2 [S,R], W [S,S], farnn, nl, semiring, gate params, sig_k."""
    farnn = p['farnn']
    if farnn == 0:
        hb = h
    else:
        z = _sig(h @ p['Wss1'] + v @ p['Wrs1'] + p['bs1'], p['sig_k'])
        if farnn == 2:
            r = _sig(h @ p['Wss2'] + v @ p['Wrs2'] + p['bs2'], p['sig_k'])
            hb = ((F32(1) - r) * h_init + r * h).astype(F32)
        else:
            hb = h
    if not fwd:
        hb = hb * o                                               # :156-157
    S1, S2, W = p['S1'], p['S2'], p['W']
    if p['semiring'] == SEMIRING_MAX:                              # :159-166
        tmp = np.einsum('br,sr->bsr', v, S1)
        Tr = np.einsum('sr,bjr->bjs', S2, tmp).astype(F32) + W
        nx = semiring_vm(hb, Tr if fwd else Tr.transpose(0, 2, 1), SEMIRING_MAX)
    elif fwd:                                                      # :169-173
        nx = ((((hb @ S1) * v) @ S2.T) + hb @ W).astype(F32)
    else:                                                          # :174-178
        nx = ((((hb @ S2) * v) @ S1.T) + hb @ W.T).astype(F32)
    if fwd:
        nx = nx * o                                                # :180-181
    nx = _nl(nx.astype(F32), p['nl'])
    if farnn == 0:
        return nx
    return ((F32(1) - z) * h + z * nx).astype(F32)                 # :195-196


def decomp_ifst_scores(p, x, lengths, P=None):
    """FARNN_S_D_W_I_S.forward_local score part, model_decompose_single.py:207-274.

    p keys: Vgen [V,R], S1,S2 [S,R], W [S,S], Cout [K,S], h0,hT [S], farnn, nl, semiring,
    and for farnn>0: Wss1,Wrs1,bs1,(Wss2,Wrs2,bs2), sig_k.
    Returns scores [B, Lmax, K] where Lmax = lengths.max() (:221).
    """
    q = {k: (np.asarray(v, F32) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    B = x.shape[0]
    L = int(np.max(lengths))
    h0, hT, Cout, Vgen = q['h0'], q['hT'], q['Cout'], q['Vgen']
    S = h0.shape[0]
    o = Cout.sum(0).astype(F32)                                    # :232 (CE1)
    xb = reverse_prefix(x, lengths)
    h0b = np.repeat(h0[None], B, 0); hTb = np.repeat(hT[None], B, 0)
    hf, hb = h0b.copy(), hTb.copy()
    fw = np.zeros((B, L + 1, S), F32); fw[:, 0] = h0
    bw = np.zeros((B, L + 1, S), F32); bw[:, 0] = hT
    for i in range(L):
        hf = decomp_ifst_step(hf, Vgen[x[:, i]], h0b, o, q, True)
        fw[:, i + 1] = hf
        hb = decomp_ifst_step(hb, Vgen[xb[:, i]], hTb, o, q, False)
        bw[:, i + 1] = hb
    rb = reverse_prefix(bw, np.asarray(lengths) + 1)
    scores = np.zeros((B, L, Cout.shape[0]), F32)
    for i in range(L):
        ab = fw[:, i + 1] * rb[:, i + 1]                           # :265-266
        scores[:, i] = np.einsum('bs,cs->bc', ab, Cout).astype(F32)   # :204
    if P is not None:
        scores = priority(scores, P)                               # :271-272
    return scores


# --------------------------------------------------------------------------- whole-path helpers
def forward_local_tags(scores, lengths, threshold, o_idx, crf_tr=None):
    """forward_local's (flat_pred) output: flatten valid positions, batch-major
    (model_onehot.py:131-146, model_decompose_single.py:276-304)."""
    if crf_tr is None:
        return decode_argmax(flatten(scores, lengths), threshold, o_idx)
    return flatten(decode_crf(scores, lengths, crf_tr, threshold, o_idx), lengths)


def onehot_crf_extension_scores(scores):
    """BASELINE config 4 (onehot + use_crf=1) is not wired in the reference
    (model_onehot.py never reads use_crf).  SURVEY.md 8a-note defines it as the composition
    the reference would produce: append two zero columns (START/STOP rows of
    C_output_mat are rand*rand_constant with rand_constant=0,
    model_decompose_single.py:78-79), then the CRF decode of model_decompose.py:349-356."""
    B, L, C = scores.shape
    return np.concatenate([scores, np.zeros((B, L, 2), F32)], axis=2)


# --------------------------------------------------------------------------- decomposed independent=1
def decomp_ind1_output_sum(p):
    """FARNN_S_D_W_I.get_output_tensor_sum, model_decompose_independent.py:210-217 (CE1: no
    wildcard_output).  Returns Osum[from, to]."""
    csum = p['Cout'].sum(0).astype(F32)                               # [R_O]
    temp = (csum[None, :] * p['S1o']).astype(F32)                     # [S,R_O]
    Tr = np.einsum('sr,jr->js', p['S2o'], temp).astype(F32)
    if p.get('Wo') is not None:
        Tr = Tr + p['Wo']
    return Tr


def decomp_ind1_step(h, v, h_init, Osum, p, fwd):
    """FARNN_S_D_W_I.get_forward_score, model_decompose_independent.py:148-197: the per-step
    transition matrix is materialised from the factors and masked by the output sum."""
    farnn = p['farnn']
    if farnn == 0:
        hb = h
    else:
        z = _sig(h @ p['Wss1'] + v @ p['Wrs1'] + p['bs1'], p['sig_k'])
        if farnn == 2:
            r = _sig(h @ p['Wss2'] + v @ p['Wrs2'] + p['bs2'], p['sig_k'])
            hb = ((F32(1) - r) * h_init + r * h).astype(F32)
        else:
            hb = h
    tmp = np.einsum('br,sr->bsr', v, p['S1'])
    Tr = np.einsum('sr,bjr->bjs', p['S2'], tmp).astype(F32)           # [b, from, to]
    Tr = (Tr + p['W']) * Osum                                          # :167-168
    nx = semiring_vm(hb, Tr if fwd else Tr.transpose(0, 2, 1), p['semiring'])
    nx = _nl(nx.astype(F32), p['nl'])
    if farnn == 0:
        return nx
    return ((F32(1) - z) * h + z * nx).astype(F32)


def decomp_ind1_scores(p, x, lengths, P=None):
    """FARNN_S_D_W_I.forward_local score part, model_decompose_independent.py:219-274.
    p keys: Vgen [V,R], S1,S2 [S,R], W [S,S], Cout [K,R_O], S1o,S2o [S,R_O], Wo (or None),
    h0,hT, farnn, nl, semiring, gate params, sig_k.  alpha = state BEFORE token i (:262)."""
    q = {k: (np.asarray(v, F32) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    B = x.shape[0]
    L = int(np.max(lengths))
    h0, hT, Vgen = q['h0'], q['hT'], q['Vgen']
    S = h0.shape[0]
    Osum = decomp_ind1_output_sum(q)
    xb = reverse_prefix(x, lengths)
    h0b = np.repeat(h0[None], B, 0); hTb = np.repeat(hT[None], B, 0)
    hf, hb = h0b.copy(), hTb.copy()
    fw = np.zeros((B, L + 1, S), F32); fw[:, 0] = h0
    bw = np.zeros((B, L + 1, S), F32); bw[:, 0] = hT
    for i in range(L):
        hf = decomp_ind1_step(hf, Vgen[x[:, i]], h0b, Osum, q, True)
        fw[:, i + 1] = hf
        hb = decomp_ind1_step(hb, Vgen[xb[:, i]], hTb, Osum, q, False)
        bw[:, i + 1] = hb
    rb = reverse_prefix(bw, np.asarray(lengths) + 1)
    s1_s2 = np.einsum('ir,jr->ijr', q['S1'], q['S2'])
    s1_s2_out = np.einsum('ir,jr->rij', q['S1o'], q['S2o'])
    K = q['Cout'].shape[0]
    scores = np.zeros((B, L, K), F32)
    for i in range(L):
        v = Vgen[x[:, i]]
        bss = np.einsum('ijr,br->bij', s1_s2, v).astype(F32) + q['W']            # :201
        ab = fw[:, i][:, :, None] * rb[:, i + 1][:, None, :]                      # :202
        abw = ab * bss                                                            # :203
        br = np.einsum('bij,rij->br', abw, s1_s2_out).astype(F32)                 # :204
        scores[:, i] = br @ q['Cout'].T                                           # :205
    if P is not None:
        scores = priority(scores, P)
    return scores


# --------------------------------------------------------------------------- decomposed independent=0
def decomp_fst_wildcard_sum(p):
    """FARNN_S_D_W.get_wildcard_tensor_origin_sum_forward, model_decompose.py:319-324.
    Returns Wsum[from, to] = sum_q (sum_c Cw[c,q]) S1w[from,q] S2w[to,q] + WW[from,to]."""
    cw = p['Cw'].sum(0).astype(F32)                                   # [R_W]
    temp = (cw[:, None] * p['S1w'].T).astype(F32)                     # 'r,sr->rs'
    res = np.einsum('sr,rj->js', p['S2w'], temp).astype(F32)
    return (res + p['WW']).astype(F32)


def decomp_fst_step(h, v, h_init, csum, Wsum, p, fwd):
    """FARNN_S_D_W.get_forward_score, model_decompose.py:243-307.  v = generalized word vector;
    the recurrence (and the gates) see _R = v * sum_c C_embed[c]  (:253)."""
    farnn = p['farnn']
    _R = (v * csum).astype(F32)
    if farnn == 0:
        hb = h
    else:
        z = _sig(h @ p['Wss1'] + _R @ p['Wrs1'] + p['bs1'], p['sig_k'])
        if farnn == 2:
            r = _sig(h @ p['Wss2'] + _R @ p['Wrs2'] + p['bs2'], p['sig_k'])
            hb = ((F32(1) - r) * h_init + r * h).astype(F32)
        else:
            hb = h
    S1, S2 = p['S1'], p['S2']
    if p['semiring'] == SEMIRING_MAX:                                  # :269-276
        tmp = np.einsum('br,sr->bsr', _R, S1)
        Tr = np.einsum('sr,bjr->bjs', S2, tmp).astype(F32) + Wsum
        nx = semiring_vm(hb, Tr if fwd else Tr.transpose(0, 2, 1), SEMIRING_MAX)
    elif fwd:                                                          # :279-284
        nx = ((((hb @ S1) * _R) @ S2.T) + hb @ Wsum).astype(F32)
    else:                                                              # :285-290
        nx = ((((hb @ S2) * _R) @ S1.T) + hb @ Wsum.T).astype(F32)
    nx = _nl(nx.astype(F32), p['nl'])
    if farnn == 0:
        return nx
    return ((F32(1) - z) * h + z * nx).astype(F32)


def decomp_fst_scores(p, x, lengths, P=None):
    """FARNN_S_D_W.forward_local score part, model_decompose.py:373-432 with get_final_score
    (:309-323).  p keys: Vgen [V,R], C [K,R], S1,S2 [S,R], Cw [K,R_W], S1w,S2w [S,R_W], WW [S,S],
    h0,hT, farnn, nl, semiring, gate params, sig_k.  alpha = state BEFORE token i (:418)."""
    q = {k: (np.asarray(v, F32) if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    B = x.shape[0]
    L = int(np.max(lengths))
    h0, hT, Vgen = q['h0'], q['hT'], q['Vgen']
    S = h0.shape[0]
    Wsum = decomp_fst_wildcard_sum(q)
    csum = q['C'].sum(0).astype(F32)                                   # :393
    xb = reverse_prefix(x, lengths)
    h0b = np.repeat(h0[None], B, 0); hTb = np.repeat(hT[None], B, 0)
    hf, hb = h0b.copy(), hTb.copy()
    fw = np.zeros((B, L + 1, S), F32); fw[:, 0] = h0
    bw = np.zeros((B, L + 1, S), F32); bw[:, 0] = hT
    for i in range(L):
        hf = decomp_fst_step(hf, Vgen[x[:, i]], h0b, csum, Wsum, q, True)
        fw[:, i + 1] = hf
        hb = decomp_fst_step(hb, Vgen[xb[:, i]], hTb, csum, Wsum, q, False)
        bw[:, i + 1] = hb
    rb = reverse_prefix(bw, np.asarray(lengths) + 1)
    K = q['C'].shape[0]
    scores = np.zeros((B, L, K), F32)
    for i in range(L):
        v = Vgen[x[:, i]]
        al, be = fw[:, i], rb[:, i + 1]
        ab = ((al @ q['S1']) * (be @ q['S2'])).astype(F32)                            # :314-316
        sc = np.einsum('bcr,br->bc', np.einsum('br,cr->bcr', v, q['C']), ab)          # :313,:317
        abw = ((al @ q['S1w']) * (be @ q['S2w'])).astype(F32)                         # :318-320
        scores[:, i] = (sc + abw @ q['Cw'].T).astype(F32)                             # :321-322
    if P is not None:
        scores = priority(scores, P)
    return scores

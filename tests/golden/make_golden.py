#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (it cannot travel to the GPU box).  It imports the reference's model /
loader / CRF classes, feeds them seeded synthetic inputs produced by
``re2nn_seq_amd.synth`` and stores inputs + the reference's outputs as small ``.npz`` /
``.json`` files.  Only data is stored -- no reference source.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixture families (SURVEY.md section 8c "fixture plan"):
  loader_*          automaton dict -> the three dfa_to_tensor_slot_*_wildcard outputs
  ifst_*            FARNN_S_O_I_S  (independent=2) x nonlinearity x semiring x priority
  fst4_*            FARNN_S_O      (independent=0)
  ind1_*            FARNN_S_O_I    (independent=1)
  decomp_*          FARNN_S_D_W_I_S x farnn{0,1,2} x crf{0,1} x additional_states x ...
  crf_*             CRF._viterbi_decode alone
  atis_ifst         ATIS-scale i-FST (V=950,S=71,C=128,B=256,L=64): seed + tags + score samples
  metrics_*         eval_seq_token / get_ner_fmeasure on a captured prediction
"""
import argparse
import io
import json
import os
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402

from re2nn_seq_amd import synth  # noqa: E402
from src_seq.farnn.model_onehot import FARNN_S_O, FARNN_S_O_I, FARNN_S_O_I_S  # noqa: E402
from src_seq.farnn.model_decompose_single import FARNN_S_D_W_I_S  # noqa: E402
from src_seq.farnn.model_decompose_independent import FARNN_S_D_W_I  # noqa: E402
from src_seq.farnn.model_decompose import FARNN_S_D_W  # noqa: E402
from src_seq.baselines.crf import CRF  # noqa: E402
from src_seq.wfa import fsa_to_tensor as ref_f2t  # noqa: E402
from src_seq.metrics.metrics import eval_seq_token, get_ner_fmeasure  # noqa: E402
from src_seq.utils import get_length_mask  # noqa: E402

torch.set_num_threads(4)


def ns(**kw):
    """The ~30 fields the reference models read from `args`."""
    d = dict(rand_constant=0.0, train_wildcard=0, train_wildcard_wildcard=0, margin=0.3,
             threshold=0.5, train_mode='sum', local_loss_func='CE1', use_priority=0,
             independent=2, update_nonlinear='none', additional_states=0, train_word_embed=0,
             use_crf=0, random=0, train_h0=0, train_hT=0, train_V_embed=0, train_c_output=1,
             farnn=0, xavier=0, bias_init=5.0, sigmoid_exponent=5, beta=1.0, train_beta=0,
             additional_nonlinear='none', random_pad_func='uniform', marryup_type='none',
             c1_kdpr=1.0, c2_kdpr=1.0, c3_pr=1.0)
    d.update(kw)
    return argparse.Namespace(**d)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', os.path.relpath(path, ROOT), os.path.getsize(path), 'bytes')


def automaton_to_json(a):
    return {'states': sorted(a['states']), 'startstate': list(a['startstate']),
            'finalstates': list(a['finalstates']),
            'transitions': {str(f): {str(t): sorted(e) for t, e in to.items()}
                            for f, to in a['transitions'].items()}}


def run_model(model, x, lengths, with_RE=True):
    """forward_score / forward_local / forward_RE of a reference onehot model."""
    xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
    label = torch.zeros_like(xt)
    with torch.no_grad():
        scores = model.forward_score(xt, label, lt, train=False).numpy()
        _, pred, _ = model.forward_local(xt, label, lt, train=False)
        out = {'scores': scores, 'flat_pred': pred.numpy()}
        if with_RE:
            pre, sre = model.forward_RE(xt, label, lt, train=False)
            out['re_pred'] = pre.numpy()
            out['re_scores'] = sre.numpy()
    return out


# ----------------------------------------------------------------------------- loaders
def gen_loader():
    dset, automaton, rules = synth.make_dataset(n_words=40, n_entity_types=3, n_states=14,
                                                seed=11, max_len=12)
    t2i = dict(dset['t2i']); t2i['<pad>'] = len(t2i)
    s2i = dset['s2i']
    meta = {'automaton': automaton_to_json(automaton), 't2i': t2i, 's2i': s2i}
    with open(os.path.join(HERE, 'loader_small.json'), 'w') as f:
        json.dump(meta, f, sort_keys=True)
    out = {}
    for ds in ('MITR-BIO', 'ATIS-BIO'):
        r = quiet(ref_f2t.dfa_to_tensor_slot_new_wildcard, automaton, t2i, s2i, dataset=ds)
        out[ds + '.new.T4'], out[ds + '.new.W4'], out[ds + '.new.WW'] = r[0], r[2], r[3]
        out[ds + '.new.final'], out[ds + '.new.start'] = r[4], r[5]
        out[ds + '.new.language'] = np.array(sorted(r[6]))
        r = quiet(ref_f2t.dfa_to_tensor_slot_independent_wildcard, automaton, t2i, s2i, dataset=ds)
        out[ds + '.ind.T'], out[ds + '.ind.W'], out[ds + '.ind.Oten'] = r[0], r[2], r[3]
        out[ds + '.ind.final'], out[ds + '.ind.start'] = r[5], r[6]
        r = quiet(ref_f2t.dfa_to_tensor_slot_single_wildcard, automaton, t2i, s2i, dataset=ds)
        out[ds + '.single.T'], out[ds + '.single.W'], out[ds + '.single.O'] = r[0], r[2], r[3]
        out[ds + '.single.Ow'] = r[4]
        out[ds + '.single.final'], out[ds + '.single.start'] = r[5], r[6]
    save('loader_small', **out)
    return dset, automaton, t2i, s2i


# ----------------------------------------------------------------------------- onehot models
def gen_onehot(dset, automaton, t2i, s2i):
    V, o_idx = len(t2i), s2i['o']
    rng = np.random.RandomState(5)
    x, lengths = synth.pad_batch(dset['query_train'][:12], 12, t2i['<pad>'])
    lengths[0] = 12 if (x[0] != t2i['<pad>']).all() else lengths[0]
    x[1, 1:] = t2i['<pad>']; lengths[1] = 1            # a length-1 row
    T, _, W, O, Ow, fin, sta, _ = quiet(ref_f2t.dfa_to_tensor_slot_single_wildcard,
                                         automaton, t2i, s2i)
    C = O.shape[0]
    pri = np.eye(len(s2i))
    for s, i in s2i.items():
        if s.startswith('b-') and ('i-' + s[2:]) in s2i:
            pri[s2i['i-' + s[2:]]][i] = -1
    cases = {}
    for nl in ('none', 'relu', 'tanh', 'relutanh'):
        for mode in ('sum', 'max'):
            for up in (0, 1):
                a = ns(update_nonlinear=nl, train_mode=mode, use_priority=up, independent=2)
                m = quiet(FARNN_S_O_I_S, T, O, W, Ow, fin, sta, pri if up else np.eye(len(s2i)),
                          a, o_idx=o_idx)
                r = run_model(m, x, lengths)
                key = '{}.{}.p{}'.format(nl, mode, up)
                for k, v in r.items():
                    cases[key + '.' + k] = v
    save('ifst_small', x=x, lengths=lengths, T=T, W=W, O=O, h0=sta, hT=fin, priority=pri,
         o_idx=np.int64(o_idx), threshold=np.float32(0.5), **cases)

    # random dense-valued i-FST (non 0/1 weights, as after training) incl. threshold 0.99 (RE.py:88)
    S = 9
    Tr = (rng.rand(V, S, S) < 0.08) * rng.rand(V, S, S); Tr[V - 1] = 0
    Wr = (rng.rand(S, S) < 0.15) * rng.rand(S, S)
    Or = (rng.rand(C, S) < 0.3) * rng.rand(C, S)
    h0 = rng.rand(S); hT = rng.rand(S)
    cases = {}
    for nl in ('none', 'tanh'):
        a = ns(update_nonlinear=nl, threshold=0.99)
        m = quiet(FARNN_S_O_I_S, Tr, Or, Wr, np.zeros(S), hT, h0, np.eye(len(s2i)), a, o_idx=o_idx)
        for k, v in run_model(m, x, lengths).items():
            cases[nl + '.' + k] = v
    save('ifst_dense', x=x, lengths=lengths, T=Tr, W=Wr, O=Or, h0=h0, hT=hT,
         o_idx=np.int64(o_idx), threshold=np.float32(0.99), **cases)

    # FST 4-D (independent=0)
    T4, _, W4, WW, fin, sta, _ = quiet(ref_f2t.dfa_to_tensor_slot_new_wildcard, automaton, t2i, s2i)
    cases = {}
    for mode in ('sum', 'max'):
        for up in (0, 1):
            a = ns(train_mode=mode, use_priority=up, independent=0)
            m = quiet(FARNN_S_O, T4, W4, WW, fin, sta, pri if up else np.eye(len(s2i)), a, o_idx=o_idx)
            for k, v in run_model(m, x, lengths).items():
                cases['{}.p{}.{}'.format(mode, up, k)] = v
    save('fst4_small', x=x, lengths=lengths, T4=T4, W4=W4, h0=sta, hT=fin, priority=pri,
         o_idx=np.int64(o_idx), threshold=np.float32(0.5), **cases)

    # independent=1
    T, _, W, Oten, _, fin, sta, _ = quiet(ref_f2t.dfa_to_tensor_slot_independent_wildcard,
                                          automaton, t2i, s2i)
    cases = {}
    for mode in ('sum', 'max'):
        for ind in (1, 2):
            a = ns(train_mode=mode, independent=ind)
            m = quiet(FARNN_S_O_I, T, Oten, W, None, fin, sta, np.eye(len(s2i)), a, o_idx=o_idx)
            for k, v in run_model(m, x, lengths).items():
                cases['{}.ind{}.{}'.format(mode, ind, k)] = v
    save('ind1_small', x=x, lengths=lengths, T=T, W=W, Oten=Oten, h0=sta, hT=fin,
         o_idx=np.int64(o_idx), threshold=np.float32(0.5), **cases)
    return x, lengths


# ----------------------------------------------------------------------------- decomposed
DECOMP_KEYS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_output_mat', 'wildcard_mat',
               'wildcard_output_vector', 'h0', 'hT', 'beta_vec', 'Wss1', 'Wrs1', 'bs1', 'Wss2',
               'Wrs2', 'bs2')


def gen_decomposed(dset, automaton, t2i, s2i, x, lengths):
    o_idx = s2i['o']
    V = len(t2i)
    rng = np.random.RandomState(9)
    t2i_nopad = {w: i for w, i in t2i.items() if w != '<pad>'}
    T, _, W, O, Ow, fin, sta, _ = quiet(ref_f2t.dfa_to_tensor_slot_single_wildcard,
                                         automaton, t2i_nopad, s2i)
    n_pairs = int((T.sum(0) > 0).sum())
    R = n_pairs + 6                  # keep R != S (pad_additional_states pads every dim == S)
    Vf, S1, S2 = synth.exact_cp_factors(T, rank=R, rng=rng, noise=0.02)
    Vf = np.append(Vf, np.zeros((1, R)), axis=0)                    # pad row (init_params.py:280-281)
    D = 8
    E = np.append(rng.randn(V - 1, D) * 0.5, np.zeros((1, D)), axis=0)
    pri = np.eye(len(s2i))
    for s, i in s2i.items():
        if s.startswith('b-') and ('i-' + s[2:]) in s2i:
            pri[s2i['i-' + s[2:]]][i] = -1
    xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
    label = torch.zeros_like(xt)
    configs = []
    for farnn in (0, 1, 2):
        for crf in (0, 1):
            configs.append(dict(farnn=farnn, use_crf=crf, update_nonlinear='tanh', beta=0.7))
    configs += [
        dict(farnn=0, use_crf=0, update_nonlinear='none', beta=1.0),
        dict(farnn=0, use_crf=0, update_nonlinear='relu', beta=0.5, additional_nonlinear='tanh'),
        dict(farnn=2, use_crf=1, update_nonlinear='relutanh', beta=0.3, additional_nonlinear='relu',
             additional_states=2, rand_constant=1e-3, use_priority=1),
        dict(farnn=1, use_crf=0, update_nonlinear='tanh', beta=0.9, additional_nonlinear='sigmoid',
             additional_states=3, rand_constant=1e-3),
        dict(farnn=0, use_crf=0, update_nonlinear='tanh', beta=0.6, train_mode='max'),
        dict(farnn=0, use_crf=1, update_nonlinear='tanh', beta=0.6, additional_nonlinear='relutanh',
             sigmoid_exponent=2),
    ]
    meta = []
    blob = {}
    for k, cfg in enumerate(configs):
        torch.manual_seed(100 + k)
        a = ns(independent=2, **cfg)
        m = quiet(FARNN_S_D_W_I_S, V=Vf, S1=S1, S2=S2, C_output_mat=O, wildcard_mat=W,
                  wildcard_output_vector=Ow, final_vector=fin, start_vector=sta,
                  pretrained_word_embed=E, priority_mat=pri, args=a, o_idx=o_idx, is_cuda=False)
        if a.use_crf:     # exercise the DP with non-default transitions
            with torch.no_grad():
                m.crf.transitions += torch.randn_like(m.crf.transitions) * 0.3
        sd = {kk: vv.detach().numpy() for kk, vv in m.state_dict().items()}
        pre = 'c{}.'.format(k)
        for kk in DECOMP_KEYS:
            if kk in sd:
                blob[pre + kk] = sd[kk]
        blob[pre + 'embedding'] = sd['embedding.weight']
        blob[pre + 'priority_mat'] = sd['priority_layer.priority_mat']
        if a.use_crf:
            blob[pre + 'crf_transitions'] = sd['crf.transitions']
        # capture scores by re-running the score part (forward_local returns only tags)
        captured = {}
        orig_decode = m.decode

        def spy(all_scores, flat, mask, lens, _c=captured, _o=orig_decode):
            _c['scores'] = all_scores.detach().numpy().copy()
            return _o(all_scores, flat, mask, lens)
        m.decode = spy
        with torch.no_grad():
            _, pred, _ = m.forward_local(xt, label, lt, train=False)
        blob[pre + 'scores'] = captured['scores']
        blob[pre + 'flat_pred'] = pred.numpy()
        meta.append(cfg)
    with open(os.path.join(HERE, 'decomp_small.json'), 'w') as f:
        json.dump({'configs': meta, 'o_idx': int(o_idx), 'threshold': 0.5}, f, sort_keys=True)
    save('decomp_small', x=x, lengths=lengths, V_in=Vf, S1_in=S1, S2_in=S2, O_in=O, W_in=W,
         Ow_in=Ow, final_in=fin, start_in=sta, E_in=E, priority_in=pri, **blob)


# ----------------------------------------------------------------------------- decomposed independent=1
IND1_KEYS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_output', 'S1_output', 'S2_output',
             'wildcard_mat', 'h0', 'hT', 'beta_vec', 'Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')


def exact_output_factors(Oten, rank, rng, noise):
    """Oten[c,s,j] = sum_q C[c,q] S1o[s,q] S2o[j,q]: one term per (s,j) pair that carries a label."""
    C, S, _ = Oten.shape
    pairs = np.argwhere(Oten.sum(0) > 0)
    assert rank >= len(pairs)
    Cf = np.zeros((C, rank)); S1o = np.zeros((S, rank)); S2o = np.zeros((S, rank))
    for q, (s, j) in enumerate(pairs):
        Cf[:, q] = Oten[:, s, j]
        S1o[s, q] = 1.0
        S2o[j, q] = 1.0
    Cf += noise * rng.randn(*Cf.shape); S1o += noise * rng.randn(*S1o.shape); S2o += noise * rng.randn(*S2o.shape)
    return Cf, S1o, S2o


def gen_decomposed_ind1(dset, automaton, t2i, s2i, x, lengths):
    o_idx = s2i['o']
    V = len(t2i)
    rng = np.random.RandomState(19)
    t2i_nopad = {w: i for w, i in t2i.items() if w != '<pad>'}
    T, _, W, Oten, _, fin, sta, _ = quiet(ref_f2t.dfa_to_tensor_slot_independent_wildcard,
                                          automaton, t2i_nopad, s2i)
    R = int((T.sum(0) > 0).sum()) + 6
    Vf, S1, S2 = synth.exact_cp_factors(T, rank=R, rng=rng, noise=0.02)
    Vf = np.append(Vf, np.zeros((1, R)), axis=0)
    RO = int((Oten.sum(0) > 0).sum()) + 3
    Cf, S1o, S2o = exact_output_factors(Oten, RO, rng, 0.02)
    D = 8
    E = np.append(rng.randn(V - 1, D) * 0.5, np.zeros((1, D)), axis=0)
    pri = np.eye(len(s2i))
    for s_, i_ in s2i.items():
        if s_.startswith('b-') and ('i-' + s_[2:]) in s2i:
            pri[s2i['i-' + s_[2:]]][i_] = -1
    xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
    label = torch.zeros_like(xt)
    configs = [
        dict(farnn=0, use_crf=0, update_nonlinear='tanh', beta=0.7),
        dict(farnn=0, use_crf=1, update_nonlinear='none', beta=1.0),
        dict(farnn=1, use_crf=0, update_nonlinear='relu', beta=0.5, additional_nonlinear='tanh'),
        dict(farnn=2, use_crf=1, update_nonlinear='tanh', beta=0.6, use_priority=1),
        dict(farnn=0, use_crf=0, update_nonlinear='tanh', beta=0.8, train_mode='max'),
        dict(farnn=2, use_crf=0, update_nonlinear='relutanh', beta=0.4, additional_states=2, rand_constant=1e-3),
    ]
    blob, meta = {}, []
    for k, cfg in enumerate(configs):
        torch.manual_seed(300 + k)
        a = ns(independent=1, **cfg)
        m = quiet(FARNN_S_D_W_I, V=Vf, S1=S1, S2=S2, C_output=Cf, S1_output=S1o, S2_output=S2o,
                  wildcard_mat=W, wildcard_output=None, final_vector=fin, start_vector=sta,
                  pretrained_word_embed=E, priority_mat=pri, args=a, o_idx=o_idx)
        m.is_cuda = False
        m.initialize()
        if a.use_crf:
            with torch.no_grad():
                m.crf.transitions += torch.randn_like(m.crf.transitions) * 0.3
        sd = {kk: vv.detach().numpy() for kk, vv in m.state_dict().items()}
        pre = 'c{}.'.format(k)
        for kk in IND1_KEYS:
            if kk in sd:
                blob[pre + kk] = sd[kk]
        blob[pre + 'embedding'] = sd['embedding.weight']
        blob[pre + 'priority_mat'] = sd['priority_layer.priority_mat']
        if a.use_crf:
            blob[pre + 'crf_transitions'] = sd['crf.transitions']
        captured = {}
        orig_decode = m.decode

        def spy(all_scores, flat, mask, lens, _c=captured, _o=orig_decode):
            _c['scores'] = all_scores.detach().numpy().copy()
            return _o(all_scores, flat, mask, lens)
        m.decode = spy
        with torch.no_grad():
            _, pred, _ = m.forward_local(xt, label, lt, train=False)
        blob[pre + 'scores'] = captured['scores']
        blob[pre + 'flat_pred'] = pred.numpy()
        meta.append(cfg)
    with open(os.path.join(HERE, 'decomp_ind1_small.json'), 'w') as f:
        json.dump({'configs': meta, 'o_idx': int(o_idx), 'threshold': 0.5}, f, sort_keys=True)
    save('decomp_ind1_small', x=x, lengths=lengths, V_in=Vf, S1_in=S1, S2_in=S2, C_in=Cf, S1o_in=S1o,
         S2o_in=S2o, W_in=W, final_in=fin, start_in=sta, E_in=E, priority_in=pri, **blob)



# ----------------------------------------------------------------------------- decomposed independent=0
FST_KEYS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_embed', 'C_wildcard', 'S1_wildcard',
            'S2_wildcard', 'wildcard_wildcard', 'h0', 'hT', 'beta_vec',
            'Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')


def exact_4d_factors(T4, rank, rng, noise):
    """T4[v,c,s,j] = sum_r V[v,r] C[c,r] S1[s,r] S2[j,r]: one term per labelled (c,s,j) edge."""
    Vn, C, S, _ = T4.shape
    trip = np.argwhere(T4.sum(0) > 0)
    assert rank >= len(trip)
    Vf = np.zeros((Vn, rank)); Cf = np.zeros((C, rank)); S1 = np.zeros((S, rank)); S2 = np.zeros((S, rank))
    for r, (c, s, j) in enumerate(trip):
        Vf[:, r] = T4[:, c, s, j]
        Cf[c, r] = 1.0; S1[s, r] = 1.0; S2[j, r] = 1.0
    for a in (Vf, Cf, S1, S2):
        a += noise * rng.randn(*a.shape)
    return Vf, Cf, S1, S2


def gen_decomposed_fst(dset, automaton, t2i, s2i, x, lengths):
    """FARNN_S_D_W (independent=0 decomposed): SURVEY.md 8a row a15, model_decompose.py:10-459."""
    o_idx = s2i['o']
    V = len(t2i)
    rng = np.random.RandomState(23)
    t2i_nopad = {w: i for w, i in t2i.items() if w != '<pad>'}
    T4, _, W4, WW, fin, sta, _ = quiet(ref_f2t.dfa_to_tensor_slot_new_wildcard, automaton, t2i_nopad, s2i)
    S = T4.shape[2]
    R = int((T4.sum(0) > 0).sum()) + 5
    RW = int((W4.sum(0) > 0).sum()) + 4
    assert S not in (R, RW, T4.shape[1])
    Vf, Cf, S1, S2 = exact_4d_factors(T4, R, rng, 0.02)
    Vf = np.append(Vf, np.zeros((1, R)), axis=0)
    Cw, S1w, S2w = exact_output_factors(W4, RW, rng, 0.02)
    D = 8
    E = np.append(rng.randn(V - 1, D) * 0.5, np.zeros((1, D)), axis=0)
    pri = np.eye(len(s2i))
    for s_, i_ in s2i.items():
        if s_.startswith('b-') and ('i-' + s_[2:]) in s2i:
            pri[s2i['i-' + s_[2:]]][i_] = -1
    xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
    label = torch.zeros_like(xt)
    configs = [
        dict(farnn=0, use_crf=0, update_nonlinear='tanh', beta=0.7),
        dict(farnn=0, use_crf=1, update_nonlinear='none', beta=1.0),
        dict(farnn=1, use_crf=0, update_nonlinear='relu', beta=0.5, additional_nonlinear='tanh'),
        dict(farnn=2, use_crf=1, update_nonlinear='tanh', beta=0.6, use_priority=1),
        dict(farnn=0, use_crf=0, update_nonlinear='tanh', beta=0.8, train_mode='max'),
        dict(farnn=2, use_crf=0, update_nonlinear='relutanh', beta=0.4, additional_states=2, rand_constant=1e-3),
    ]
    blob, meta = {}, []
    for k, cfg in enumerate(configs):
        torch.manual_seed(400 + k)
        a = ns(independent=0, **cfg)
        m = quiet(FARNN_S_D_W, V=Vf, C=Cf, S1=S1, S2=S2, C_wildcard=Cw, S1_wildcard=S1w, S2_wildcard=S2w,
                  wildcard_wildcard=WW, final_vector=fin, start_vector=sta, pretrained_word_embed=E,
                  priority_mat=pri, args=a, o_idx=o_idx)
        m.is_cuda = False
        m.initialize()
        if a.use_crf:
            with torch.no_grad():
                m.crf.transitions += torch.randn_like(m.crf.transitions) * 0.3
        sd = {kk: vv.detach().numpy() for kk, vv in m.state_dict().items()}
        pre = 'c{}.'.format(k)
        for kk in FST_KEYS:
            if kk in sd:
                blob[pre + kk] = sd[kk]
        blob[pre + 'embedding'] = sd['embedding.weight']
        blob[pre + 'priority_mat'] = sd['priority_layer.priority_mat']
        if a.use_crf:
            blob[pre + 'crf_transitions'] = sd['crf.transitions']
        captured = {}
        orig_decode = m.decode

        def spy(all_scores, flat, mask, lens, _c=captured, _o=orig_decode):
            _c['scores'] = all_scores.detach().numpy().copy()
            return _o(all_scores, flat, mask, lens)
        m.decode = spy
        with torch.no_grad():
            _, pred, _ = m.forward_local(xt, label, lt, train=False)
        blob[pre + 'scores'] = captured['scores']
        blob[pre + 'flat_pred'] = pred.numpy()
        meta.append(cfg)
    with open(os.path.join(HERE, 'decomp_fst_small.json'), 'w') as f:
        json.dump({'configs': meta, 'o_idx': int(o_idx), 'threshold': 0.5}, f, sort_keys=True)
    save('decomp_fst_small', x=x, lengths=lengths, V_in=Vf, C_in=Cf, S1_in=S1, S2_in=S2, Cw_in=Cw,
         S1w_in=S1w, S2w_in=S2w, WW_in=WW, final_in=fin, start_in=sta, E_in=E, priority_in=pri, **blob)


# ----------------------------------------------------------------------------- CRF alone
def gen_crf():
    rng = np.random.RandomState(21)
    B, L, K = 9, 11, 7
    feats = rng.randn(B, L, K).astype(np.float32)
    feats[0] = np.round(feats[0])             # exact ties
    feats[1] = 0.0
    lengths = rng.randint(1, L + 1, size=B).astype(np.int64)
    lengths[2] = L; lengths[3] = 1
    crf = quiet(CRF, K - 2, False)
    out = {}
    with torch.no_grad():
        mask = get_length_mask(torch.from_numpy(lengths), L)
        _, p = crf._viterbi_decode(torch.from_numpy(feats), mask)
        out['default.paths'] = p.numpy().copy()
        out['default.tr'] = crf.transitions.detach().numpy().copy()
        crf.transitions += torch.from_numpy(rng.randn(K, K).astype(np.float32))
        _, p = crf._viterbi_decode(torch.from_numpy(feats), mask)
        out['random.paths'] = p.numpy().copy()
        out['random.tr'] = crf.transitions.detach().numpy().copy()
    save('crf_small', feats=feats, lengths=lengths, **out)


# ----------------------------------------------------------------------------- ATIS scale
def gen_atis_scale():
    seed = 1234
    rng = np.random.RandomState(seed)
    V, S, C, B, L = 950, 71, 128, 256, 64
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
    x, lengths = synth.random_batch(V, B, L, rng)
    # plant accepting paths so that rules fire: walk the automaton from state 0
    succ = {}
    ws, ss, js = np.nonzero(T)
    for w, s, j in zip(ws, ss, js):
        succ.setdefault(int(s), []).append((int(w), int(j)))
    for b in range(0, B, 2):
        n = int(lengths[b]); at = int(rng.randint(0, max(1, n - 4))); s = 0
        for t in range(at, n):
            if s not in succ:
                break
            w, s = succ[s][int(rng.randint(len(succ[s])))]
            x[b, t] = w
    a = ns(independent=2)
    m = quiet(FARNN_S_O_I_S, T.astype(np.float64), O.astype(np.float64), W.astype(np.float64),
              np.zeros(S), hT.astype(np.float64), h0.astype(np.float64), np.eye(C - 1), a, o_idx=0)
    r = run_model(m, x, lengths, with_RE=True)
    rows = np.array([0, 1, 2, 3, 100, 101, 254, 255])
    save('atis_ifst', seed=np.int64(seed), x=x.astype(np.int16), lengths=lengths.astype(np.int16),
         tags=r['re_pred'].astype(np.int16), flat_pred=r['flat_pred'].astype(np.int16),
         sample_rows=rows, sample_scores=r['scores'][rows].astype(np.float32),
         dims=np.array([V, S, C, B, L]))
    n_fired = int((r['flat_pred'] != 0).sum())
    print('atis_ifst: valid tokens', int(lengths.sum()), 'non-O tags', n_fired)


# ----------------------------------------------------------------------------- metrics
def gen_metrics(dset, s2i):
    rng = np.random.RandomState(3)
    i2s = dset['i2s']
    n = 400
    true = rng.randint(0, len(s2i), size=n)
    pred = np.where(rng.rand(n) < 0.7, true, rng.randint(0, len(s2i), size=n))
    tl = [torch.tensor(int(v)) for v in true]
    pl = [torch.tensor(int(v)) for v in pred]
    acc, p, r, f = eval_seq_token(seq_label_pred=pl, seq_label_true=tl, o_idx=s2i['o'])
    a2, p2, r2, f2, cls = quiet(get_ner_fmeasure, golden_lists=tl, predict_lists=pl,
                                label_type="BIO", i2s=i2s, all_class=True)
    with open(os.path.join(HERE, 'metrics_small.json'), 'w') as fo:
        json.dump({'true': true.tolist(), 'pred': pred.tolist(), 'o_idx': int(s2i['o']),
                   'i2s': {str(k): v for k, v in i2s.items()},
                   'token': [float(acc), float(p), float(r), float(f)],
                   'entity': [float(a2), float(p2), float(r2), float(f2)],
                   'per_class': {k: [float(t) for t in v] for k, v in cls.items()}}, fo, sort_keys=True)
    print('wrote tests/golden/metrics_small.json')


def gen_cli_flags():
    """Flag names / types / defaults of the reference CLI, parsed from its argparse calls (the
    module itself cannot be imported here: it pulls in fasttext/pydash via data.py).  Also copies
    one of the reference's shipped example result files (a data file) as the --args_path fixture."""
    import re
    import shutil
    src = open('/root/reference/src_seq/main.py').read()
    flags = []
    for m in re.finditer(r"parser\.add_argument\('--(\w+)',\s*type=(\w+),\s*default=([^,]+?),\s*help", src):
        flags.append([m.group(1), m.group(2), eval(m.group(3).strip())])
    with open(os.path.join(HERE, 'cli_flags.json'), 'w') as f:
        json.dump({'flags': flags}, f, indent=0)
    shutil.copy('/root/reference/model_seq/example/ATIS-ZH.FSTRNN.0%.softmax.7503.res',
                os.path.join(HERE, 'example_ATIS-ZH_0pct.res'))
    print('wrote tests/golden/cli_flags.json ({} flags) and example_ATIS-ZH_0pct.res'.format(len(flags)))



if __name__ == '__main__':
    dset, automaton, t2i, s2i = gen_loader()
    x, lengths = gen_onehot(dset, automaton, t2i, s2i)
    gen_decomposed(dset, automaton, t2i, s2i, x, lengths)
    gen_decomposed_ind1(dset, automaton, t2i, s2i, x, lengths)
    gen_decomposed_fst(dset, automaton, t2i, s2i, x, lengths)
    gen_crf()
    gen_atis_scale()
    gen_metrics(dset, s2i)
    gen_cli_flags()

"""The regular-expression teacher: run the ONEHOT tagger over train/dev/test and keep the
unflattened predictions and scores (reference src_seq/RE.py:15-51 get_RE_prediction, :77-192
predict_by_RE).  This is the second production caller of the hot path: it needs every position
of every padded row (pads are run through the recurrence like the reference's padded loop) and
the full score tensor, i.e. ``forward_RE`` / FARNN_MODE_FULL.

Result (also cached next to the automaton as ``<automata_path>.re.score``, ref :190):
    (pred_train[N,L], pred_dev, pred_test, score_train[N,L,C], score_dev, score_test)
"""
import os
import pickle
from copy import copy

import torch

from .create_logic_mat_bias import create_mat_priority_MITR
from .data import SlotBatchDatasetNoRE, iter_batches, load_slot_dataset
from .farnn.model_onehot import FARNN_S_O, FARNN_S_O_I, FARNN_S_O_I_S
from .metrics.metrics import eval_seq_token, get_ner_fmeasure
from .utils import load_pkl, pad_dataset_1, set_seed
from .wfa import fsa_to_tensor as f2t

# the reference hard-codes one automaton per dataset (:53-64); override with args.re_automata_path
DEFAULT_RE_AUTOMATA = {
    'ATIS-BIO': '../data/ATIS-BIO/automata/automata.INTEGRATE.1025152019-1603639219.365244.150.'
                '100:0.3857,150:0.3004,.seed|168|.random.3best.71states.1splits.pkl',
    'ATIS-ZH-BIO': '../data/ATIS-ZH-BIO/automata/IIID.automata.0308133803-1615210683.6103313.svd.2best.'
                   '104states.random.3splits.100-0.1496150-0.0934200-0.0541.bio.rules.v1.config.pkl',
    'SNIPS-BIO': '../data/SNIPS-BIO/automata/IIID.automata.0323152125-1616512885.4562736.svd.2best.'
                 '104states.random.1splits.200-0.0025250-0.0026300-0.0022.bio.rules.v1.config.pkl',
}


def get_RE_prediction(dataloader, model, args, o_idx=0, i2s=None):
    preds, scores, flat_pred, flat_true = [], [], [], []
    model.eval()
    with torch.no_grad():
        for batch in dataloader:
            x, label, lengths = batch['x'], batch['s'], batch['l']
            pred_label, all_scores = model.forward_RE(x, label, lengths, train=False)
            pred_cpu = pred_label.cpu()
            preds.append(pred_cpu)
            scores.append(all_scores.cpu())
            flat_pred.append(model._flatten(pred_cpu, lengths))      # numpy-backed on host tensors
            flat_true.append(model._flatten(label, lengths))
    fp, ft = torch.cat(flat_pred), torch.cat(flat_true)
    acc, p, r, f = eval_seq_token(seq_label_pred=fp, seq_label_true=ft, o_idx=o_idx)
    acc_ner, p_ner, r_ner, f_ner, class_res = get_ner_fmeasure(
        golden_lists=ft, predict_lists=fp, label_type="BIO", i2s=i2s, all_class=True)
    print({'token-level': [acc, p, r, f], 'entity-level': [acc_ner, p_ner, r_ner, f_ner, class_res]})
    pred_all = torch.cat(preds, dim=0)
    score_all = torch.cat(scores, dim=0)
    # the clamped `oo` score (threshold 0.99, :88) is restored to 1.0 (:48); the reference compares
    # against the Python double 0.99, so the float32 image of 0.99 is matched explicitly
    score_all[score_all == torch.tensor(0.99, dtype=torch.float32)] = 1.0
    model.train()
    return pred_all, score_all


def assign_automata(args_bak):
    override = getattr(args_bak, 're_automata_path', None)
    if override:
        args_bak.automata_path = override
    elif args_bak.dataset in DEFAULT_RE_AUTOMATA:
        args_bak.automata_path = DEFAULT_RE_AUTOMATA[args_bak.dataset]
    else:
        raise NotImplementedError(args_bak.dataset)
    return args_bak


def build_onehot_model(args, automata, t2i, s2i, priority_mat):
    """Automaton dict -> onehot tagger for args.independent (reference train_onehot.py:80-127)."""
    if args.local_loss_func != 'CE1':
        raise NotImplementedError('only CE1 is reachable from main.py (:127)')
    o_idx = s2i['o']
    cls = {1: FARNN_S_O_I, 2: FARNN_S_O_I_S}.get(args.independent, FARNN_S_O)
    if not args.rand_constant and not os.environ.get('FARNN_DENSE_LOADER'):
        # no noise to add: skip the dense float64 host tensors and scatter the edges in HBM;
        # same default dataset kwarg as the dense calls below (the 'MITR-BIO' quirk)
        return cls.from_automaton(automata, t2i, s2i, priority_mat, args, o_idx=o_idx)
    if args.independent == 1:
        T, _, W, Oten, Ow, fin, sta, _ = f2t.dfa_to_tensor_slot_independent_wildcard(automata, t2i, s2i)
        return FARNN_S_O_I(T, Oten, W, Ow, fin, sta, priority_mat, args, o_idx=o_idx)
    if args.independent == 2:
        T, _, W, O, Ow, fin, sta, _ = f2t.dfa_to_tensor_slot_single_wildcard(automata, t2i, s2i)
        return FARNN_S_O_I_S(T, O, W, Ow, fin, sta, priority_mat, args, o_idx=o_idx)
    T4, _, W4, WW, fin, sta, _ = f2t.dfa_to_tensor_slot_new_wildcard(automata, t2i, s2i)
    return FARNN_S_O(T4, W4, WW, fin, sta, priority_mat, args, o_idx=o_idx)


def predict_by_RE(args, data_dir='../data/'):
    args_bak = copy(args)
    args_bak.data_type = 'all'          # (:85-90) the teacher's fixed settings
    args_bak.beta = 1
    args_bak.threshold = 0.99
    args_bak.rand_constant = 0
    args_bak.use_crf = 0
    args_bak = assign_automata(args_bak)
    set_seed(args.seed)
    cache = args_bak.automata_path + '.re.score'
    if os.path.exists(cache):
        with open(cache, 'rb') as f:
            return pickle.load(f)

    dset = load_slot_dataset(args.dataset, data_dir)
    t2i, i2t, s2i, i2s = dset['t2i'], dset['i2t'], dset['s2i'], dset['i2s']
    if '<pad>' not in t2i:
        i2t[len(i2t)] = '<pad>'
        t2i['<pad>'] = len(i2t) - 1
    L, pad = args.seq_max_len, t2i['<pad>']
    loaders = []
    for name in ('train', 'dev', 'test'):
        q, _, lens = pad_dataset_1(dset['query_' + name], L, pad)
        s, _, _ = pad_dataset_1(dset['intent_' + name], L, s2i['o'])
        loaders.append(SlotBatchDatasetNoRE(q, lens, s, args_bak, s2i))
    automata = load_pkl(args_bak.automata_path)
    if 'automata' in automata:
        automata = automata['automata']
    print("AUTOMATA STATES NUM: {}".format(len(automata['states'])))
    model = build_onehot_model(args_bak, automata, t2i, s2i, create_mat_priority_MITR(s2i))
    outs = [get_RE_prediction(iter_batches(d, args.bz), model, args_bak, s2i['o'], i2s) for d in loaders]
    result = (outs[0][0], outs[1][0], outs[2][0], outs[0][1], outs[1][1], outs[2][1])
    with open(cache, 'wb') as f:
        pickle.dump(result, f)
    return result

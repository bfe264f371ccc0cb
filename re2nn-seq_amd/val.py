"""Evaluation loop over a dataset: the caller of the hot path (reference src_seq/val.py:7-43).

Same signature and result dict as the reference's ``val_onehot``.  The reference appends every
predicted token to a Python list as a 0-d tensor and scores them in pure-Python loops; here the
flat predictions stay tensors until one concatenation, and the metrics are vectorised."""
import collections
import time

import torch

from .data import iter_batches
from .metrics.metrics import eval_seq_token, get_ner_fmeasure


def val_onehot(dataloader, model, args, o_idx=0, i2s=None, i2t=None, is_cuda=True, stats=None):
    """`dataloader`: any iterable of {'x','s','l'} batches (a torch DataLoader, or a dataset from
    data.py which is then batched with args.bz).  `stats` (optional dict) receives timing."""
    if hasattr(dataloader, '__getitem__') and not hasattr(dataloader, '__iter__'):
        dataloader = iter_batches(dataloader, args.bz)
    preds, trues = [], []
    n_tok, t0 = 0, time.perf_counter()
    model.eval()
    # the reference calls forward_local and reads the result batch by batch (val.py:28-31); here up to
    # `pipeline_depth` batches are in flight: the H2D copy and launch of batch i+1 are enqueued before the
    # predictions of batch i-1 are read back (same calls, same order of results)
    submit = getattr(model, 'submit_local', None)
    depth = getattr(model, 'pipeline_depth', 0) if submit is not None and not getattr(model, 'training', False) else 0
    pending = collections.deque()

    def collect(item):
        _, pred_label, true_label = item.result()
        preds.append(pred_label.reshape(-1).cpu())
        trues.append(true_label.reshape(-1).cpu())

    with torch.no_grad():
        for batch in dataloader:
            x, label, lengths = batch['x'], batch['s'], batch['l']
            if depth > 0:
                pending.append(submit(x, label, lengths))
                if len(pending) > depth:
                    collect(pending.popleft())
            else:
                _, pred_label, true_label = model.forward_local(x, label, lengths, train=False)
                preds.append(pred_label.reshape(-1).cpu())
                trues.append(true_label.reshape(-1).cpu())
            n_tok += int(lengths.sum())
        while pending:
            collect(pending.popleft())
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    all_pred = torch.cat(preds) if preds else torch.zeros(0, dtype=torch.int64)
    all_true = torch.cat(trues) if trues else torch.zeros(0, dtype=torch.int64)
    acc, p, r, f = eval_seq_token(seq_label_pred=all_pred, seq_label_true=all_true, o_idx=o_idx)
    acc_ner, p_ner, r_ner, f_ner, class_res = get_ner_fmeasure(
        golden_lists=all_true, predict_lists=all_pred, label_type="BIO", i2s=i2s, all_class=True)
    if stats is not None:
        stats.update(tokens=n_tok, seconds=elapsed, tokens_per_s=n_tok / elapsed if elapsed > 0 else 0.0)
    model.train()
    return {'token-level': [acc, p, r, f],
            'entity-level': [acc_ner, p_ner, r_ner, f_ner, class_res]}

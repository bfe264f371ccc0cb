"""Token-level and entity-level (BIO span) scores of a flat prediction, with the same return
tuples as the reference's src_seq/metrics/metrics.py (:7-42 eval_seq_token, :96-129
get_ner_fmeasure, :184-229 get_ner_BIO).  The reference walks Python lists of 0-d tensors one
token at a time; here the token metrics are numpy reductions and the span extraction is a single
pass over label strings."""
import numpy as np


def _as_int_array(seq):
    if hasattr(seq, 'detach'):
        return seq.detach().cpu().numpy().astype(np.int64).ravel()
    return np.asarray([int(v) for v in seq], dtype=np.int64) if not isinstance(seq, np.ndarray) \
        else seq.astype(np.int64).ravel()


def eval_seq_token(seq_label_pred, seq_label_true, o_idx=0):
    pred, true = _as_int_array(seq_label_pred), _as_int_array(seq_label_true)
    assert len(pred) == len(true)
    same = pred == true
    tp = int(np.sum(same & (pred != o_idx)))
    fp = int(np.sum(~same & (pred != o_idx)))
    fn = int(np.sum(~same & (true != o_idx)))
    accuracy = float(np.sum(same)) / len(pred)
    precision = tp / (tp + fp) if (tp + fp) else 0
    recall = tp / (tp + fn) if (tp + fn) else 0
    f1 = 2 * precision * recall / (precision + recall) if (precision + recall) else 0
    return accuracy, precision, recall, f1


def get_ner_BIO(label_list):
    """Spans as '[start,end]TYPE' strings (ref :184-229 incl. its quirks: an I- tag of another
    type closes the open span and is itself dropped; the last open span has no end index)."""
    spans = []
    open_span, open_type = '', ''
    for i, raw in enumerate(label_list):
        lab = raw.upper()
        if 'B-' in lab:
            if open_type != '':
                spans.append(open_span + ',' + str(i - 1))
            open_type = lab.replace('B-', '', 1)
            open_span = open_type + '[' + str(i)
        elif 'I-' in lab:
            if lab.replace('I-', '', 1) != open_type:
                if open_span != '' and open_type != '':
                    spans.append(open_span + ',' + str(i - 1))
                open_span, open_type = '', ''
        else:
            if open_span != '' and open_type != '':
                spans.append(open_span + ',' + str(i - 1))
            open_span, open_type = '', ''
    if open_span != '' and open_type != '':
        spans.append(open_span)
    out = []
    for s in spans:
        if len(s) > 0:
            s = s + ']'
            k = s.index('[')
            out.append(s[k:] + s[:k])
    return out


def _prf(pred_spans, gold_spans):
    right = len(set(gold_spans).intersection(set(pred_spans)))
    precision = right / len(pred_spans) if len(pred_spans) else -1
    recall = right / len(gold_spans) if len(gold_spans) else -1
    if precision == -1 or recall == -1 or (precision + recall) <= 0.:
        return precision, recall, -1
    return precision, recall, 2 * precision * recall / (precision + recall)


def get_ner_fmeasure(golden_lists, predict_lists, label_type="BIO", i2s=None, all_class=False):
    if label_type in ("BMES", "BIOES"):
        raise NotImplementedError('only BIO datasets are reachable from main.py (:180)')
    gold = [i2s[int(m)] for m in _as_int_array(golden_lists)]
    pred = [i2s[int(m)] for m in _as_int_array(predict_lists)]
    accuracy = sum(1 for a, b in zip(gold, pred) if a == b) / len(gold)
    gold_spans, pred_spans = get_ner_BIO(gold), get_ner_BIO(pred)
    precision, recall, f_measure = _prf(pred_spans, gold_spans)
    per_class = None
    if all_class:
        buckets = {}
        for s in pred_spans:
            buckets.setdefault(s.split(']')[1], [[], []])[0].append(s)
        for s in gold_spans:
            buckets.setdefault(s.split(']')[1], [[], []])[1].append(s)
        per_class = {k: list(_prf(v[0], v[1])) for k, v in buckets.items()}
    return accuracy, precision, recall, f_measure, per_class

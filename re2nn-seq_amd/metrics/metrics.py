"""Token-level and entity-level (BIO span) scores of a flat prediction, with the same return
tuples as the reference's src_seq/metrics/metrics.py (:7-42 eval_seq_token, :96-129
get_ner_fmeasure, :184-229 get_ner_BIO).  The reference walks Python lists of 0-d tensors one
token at a time; here the token metrics are numpy reductions and the entity-level scores come from
a vectorised restatement of the span automaton (bio_spans); get_ner_BIO keeps the reference's
string form for callers that want the spans themselves."""
import numpy as np


def _as_int_array(seq):
    if hasattr(seq, 'detach'):
        return seq.detach().cpu().numpy().astype(np.int64, copy=False).ravel()
    return np.asarray([int(v) for v in seq], dtype=np.int64) if not isinstance(seq, np.ndarray) \
        else seq.astype(np.int64, copy=False).ravel()


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


def _label_tables(i2s):
    """Per label id: kind (0 other, 1 B-, 2 I-) and an integer type, using the reference's string
    operations on the upper-cased label (:190-216).  Returns None when a B- label has an empty type
    (the reference then treats the span as not open: left to the string path)."""
    n = max(int(k) for k in i2s) + 1
    kind = np.zeros(n, np.int64)
    typ = np.full(n, -1, np.int64)
    names = {}
    for k, raw in i2s.items():
        lab = raw.upper()
        if 'B-' in lab:
            t = lab.replace('B-', '', 1)
            if t == '':
                return None
            kind[int(k)] = 1
        elif 'I-' in lab:
            t = lab.replace('I-', '', 1)
            kind[int(k)] = 2
        else:
            continue
        typ[int(k)] = names.setdefault(t, len(names))
    return kind, typ, list(names)


def bio_spans(ids, kind_of, type_of):
    """Spans of a flat id sequence as arrays (start, end, type); end = -1 for the span left open at
    the end of the list.  Same automaton as get_ner_BIO: a span opens at every B-, runs through the
    I- tags of its type that follow without a gap, and anything else closes it (an I- of another
    type closes it and is dropped).  Position j continues a span iff it is I-, its left neighbour is
    B-/I- of the same type, and the chain of such links reaches back to a B-."""
    ids = np.asarray(ids, np.int64)
    n = len(ids)
    if n == 0:
        z = np.zeros(0, np.int64)
        return z, z, z
    kind, typ = kind_of[ids], type_of[ids]
    link = np.zeros(n, bool)
    link[1:] = (kind[1:] == 2) & (kind[:-1] != 0) & (typ[1:] == typ[:-1])
    starts = np.nonzero(kind == 1)[0]
    stops = np.nonzero(~link)[0]                       # every span start is one of these
    nxt = np.searchsorted(stops, starts, side='right')
    ends = np.where(nxt < len(stops), stops[np.minimum(nxt, len(stops) - 1)] - 1, -1)
    return starts, ends, typ[starts]


def _span_keys(starts, ends, types, n, ntypes):
    return (starts * (n + 1) + (ends + 1)) * max(ntypes, 1) + types


def get_ner_fmeasure(golden_lists, predict_lists, label_type="BIO", i2s=None, all_class=False):
    if label_type in ("BMES", "BIOES"):
        raise NotImplementedError('only BIO datasets are reachable from main.py (:180)')
    gold_ids, pred_ids = _as_int_array(golden_lists), _as_int_array(predict_lists)
    tables = _label_tables(i2s)
    if tables is None:
        return _get_ner_fmeasure_strings(gold_ids, pred_ids, i2s, all_class)
    kind_of, type_of, names = tables
    n = len(gold_ids)
    # the reference compares label strings (:103-107): two ids with one spelling count as equal
    spell = {}
    canon = np.arange(len(kind_of))
    for k, raw in i2s.items():
        canon[int(k)] = spell.setdefault(raw, int(k))
    accuracy = float(np.sum(canon[gold_ids] == canon[pred_ids])) / n
    gs, ge, gt = bio_spans(gold_ids, kind_of, type_of)
    ps, pe, pt = bio_spans(pred_ids, kind_of, type_of)
    gk, pk = _span_keys(gs, ge, gt, n, len(names)), _span_keys(ps, pe, pt, n, len(names))
    hit = np.isin(pk, gk, assume_unique=True)          # starts are unique, so keys are

    def prf(right, n_pred, n_gold):
        precision = right / n_pred if n_pred else -1
        recall = right / n_gold if n_gold else -1
        if precision == -1 or recall == -1 or (precision + recall) <= 0.:
            return precision, recall, -1
        return precision, recall, 2 * precision * recall / (precision + recall)

    precision, recall, f_measure = prf(int(hit.sum()), len(pk), len(gk))
    per_class = None
    if all_class:
        nt = len(names)
        right_c = np.bincount(pt[hit], minlength=nt)
        pred_c, gold_c = np.bincount(pt, minlength=nt), np.bincount(gt, minlength=nt)
        u, first = np.unique(np.concatenate([pt, gt]), return_index=True)
        order = [int(t) for t in u[np.argsort(first)]]                   # first appearance, as the reference's dict
        per_class = {names[t]: list(prf(int(right_c[t]), int(pred_c[t]), int(gold_c[t]))) for t in order}
    return accuracy, precision, recall, f_measure, per_class


def _get_ner_fmeasure_strings(gold_ids, pred_ids, i2s, all_class):
    gold = [i2s[int(m)] for m in gold_ids]
    pred = [i2s[int(m)] for m in pred_ids]
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

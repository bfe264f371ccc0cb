"""Host-side helpers of the tagging drivers (reference src_seq/utils.py, the parts the forward
tagging path touches: :28-56 padding, :112-122 Logger, :146-150 load_pkl, :202-225 get_average,
:247-261 sample selection).  The tensor helpers of that file (reverse/flatten/_matmul/_maxmul,
:153-199) have no counterpart here: they are fused into the HIP kernels."""
import datetime
import os
import pickle
import random
import time

import numpy as np


def set_seed(seed):
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def pad_dataset_1(query, seq_max_len, pad_idx):
    """Truncate / right-pad every non-empty sentence to `seq_max_len` (ref :28-56).
    Returns (padded, padded_reversed, lengths) like the reference."""
    out, out_rev, lengths = [], [], []
    for q in query:
        n = len(q)
        if n <= 0:
            continue
        q = np.asarray(q)
        rev = q[::-1]
        if n > seq_max_len:
            q, rev, n = q[:seq_max_len], rev[:seq_max_len], seq_max_len
        else:
            fill = np.repeat(pad_idx, seq_max_len - n)
            q, rev = np.concatenate((q, fill)), np.concatenate((rev, fill))
        out.append(q)
        out_rev.append(rev)
        lengths.append(n)
    return out, out_rev, lengths


def len_stats(query):
    lens = [len(q) for q in query]
    print("max_len: {}, avg_len: {}".format(max(lens), sum(lens) / len(lens)))


def load_pkl(path):
    print(path)
    with open(path, 'rb') as f:
        return pickle.load(f)


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def create_datetime_str():
    return datetime.datetime.today().strftime("%m%d%H%M%S") + '-' + str(time.time())


class Logger:
    """ref :112-122 -- a list of strings that ends up inside the .res pickle."""

    def __init__(self):
        self.record = []

    def add(self, string):
        assert type(string) == str
        self.record.append(string + ' \n')

    def save(self, filename):
        with open(filename, 'w', encoding='utf-8') as f:
            f.writelines(self.record)


def get_average(M, normalize_type):
    """Averaged norm used by --normalize_automata (ref :202-225)."""
    assert normalize_type in ['l1', 'l2', 'l1-rank', 'l2-rank']
    order = 1 if normalize_type.startswith('l1') else 2
    if normalize_type.endswith('rank'):
        return np.linalg.norm(M, order, axis=0) / M.shape[0]
    return np.linalg.norm(M, order) / M.size


def even_select_from_total_number(L, N, seed=0):
    """ref :247-261 (despite the name it is a random choice without replacement)."""
    if 0 < N < L:
        return np.random.choice(L, N, replace=False)
    if N >= L:
        return np.arange(L)
    return np.array([], dtype=np.int64)


def xavier_normal(obj):
    std = np.sqrt(2. / np.sum(obj.shape))
    return np.random.normal(loc=0., scale=std, size=obj.shape)

"""Dataset plumbing of the tagging drivers (reference src_seq/data.py: :85-96 embedding loaders,
:119-154 SlotBatchDatasetNoRE, :158-212 SlotBatchDataset, :339-341 load_slot_dataset).

On-disk schema of ``dataset.pkl`` (writer: reference data.py:412-418): dict with t2i/i2t/s2i/i2s
and query_{train,dev,test} / intent_{train,dev,test} (the "intent" lists hold SLOT ids).
The preprocessing half of the reference file (raw readers, glove/fasttext builders) is offline
tooling and out of scope.
"""
import os
import pickle

import numpy as np

from .utils import even_select_from_total_number

DATASETS = ('ATIS-BIO', 'ATIS-ZH-BIO', 'SNIPS-BIO')


def load_slot_dataset(dataset, datadir='../data/'):
    assert dataset in DATASETS
    with open(os.path.join(datadir, dataset, 'dataset.pkl'), 'rb') as f:
        return pickle.load(f)


def load_glove_embed(dataset_path, embed_dim):
    with open(os.path.join(dataset_path, 'glove.{}.emb'.format(embed_dim)), 'rb') as f:
        return pickle.load(f)


def load_fasttext_embed(dataset_path, embed_dim):
    with open(os.path.join(dataset_path, 'fasttext.{}.emb'.format(embed_dim)), 'rb') as f:
        return pickle.load(f)


def _subset(n, portion, dset, args):
    """Indices kept for `portion` (ref :126-139); None = keep everything."""
    if portion == 1.0 or portion == 0.0:
        return None
    size = int(portion) if portion > 1 else int(portion * n)     # >1 means "shots"
    if dset == 'dev':
        size = max(size, 200)                                     # dev keeps at least 200 samples
    return even_select_from_total_number(n, size, seed=args.seed)


class SlotBatchDatasetNoRE:
    """Indexable dataset of {'x','s','l'} int64 arrays (ref :119-154); works with
    torch.utils.data.DataLoader's default collate and with `iter_batches` below."""

    def __init__(self, query, lengths, slot, args, s2i, portion=1, dset='train', re_scores=None):
        assert dset in ['train', 'dev', 'test']
        assert len(query) == len(slot)
        idxs = _subset(len(query), portion, dset, args)
        if idxs is None:
            self.dataset, self.slot, self.lengths = query, slot, lengths
            self.re = None if re_scores is None else list(np.asarray(re_scores))
        else:
            self.dataset = list(np.array(query)[idxs])
            self.slot = list(np.array(slot)[idxs])
            self.lengths = list(np.array(lengths)[idxs])
            self.re = None if re_scores is None else list(np.asarray(re_scores)[idxs])

    def __getitem__(self, idx):
        item = {'x': np.array(self.dataset[idx], dtype=np.int64),
                's': np.array(self.slot[idx], dtype=np.int64),
                'l': np.array(self.lengths[idx], dtype=np.int64)}
        if self.re is not None:
            item['re'] = np.array(self.re[idx], dtype=np.float32)
        return item

    def __len__(self):
        return len(self.dataset)


class SlotBatchDataset(SlotBatchDatasetNoRE):
    """ref :158-212.  The reference runs the regular-expression teacher (predict_by_RE) inside
    this constructor to attach its scores as 're'; they feed only the KD/PR training losses and
    --use_unlabel, so on the forward tagging path the teacher is run on demand: `with_re=True`
    (implied by `--marryup_type kd|pr` or `--use_unlabel 1`)."""

    def __init__(self, query, lengths, slot, args, s2i, portion=1, dset='train', with_re=None,
                 data_dir='../data/'):
        if with_re is None:
            with_re = bool(getattr(args, 'use_unlabel', 0)) or \
                getattr(args, 'marryup_type', 'none') in ('kd', 'pr')
        re_out = None
        if with_re:
            from .RE import predict_by_RE
            preds = predict_by_RE(args, data_dir=data_dir)
            k = {'train': 0, 'dev': 1, 'test': 2}[dset]
            re_pred, re_out = preds[k], preds[3 + k]
            if args.use_unlabel and dset != 'test':
                slot = [np.asarray(re_pred[i]) for i in range(len(re_pred))]
        super().__init__(query, lengths, slot, args, s2i, portion, dset, re_scores=re_out)


def iter_batches(dataset, batch_size):
    """Minimal DataLoader(batch_size=bz) (no shuffle, no workers: reference train_onehot.py:64-66)
    yielding dicts of torch tensors."""
    import torch
    n = len(dataset)
    for lo in range(0, n, batch_size):
        items = [dataset[i] for i in range(lo, min(lo + batch_size, n))]
        yield {k: torch.from_numpy(np.stack([it[k] for it in items])) for k in items[0]}

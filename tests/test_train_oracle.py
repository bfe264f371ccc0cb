"""The training-step oracle (torch fp32 + autograd on the CPU) against loss and gradients captured from the
reference's own forward_local(train=True) + backward (tests/golden/make_golden_train.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

from oracle import farnn_train_oracle as to  # noqa: E402

GATES = ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')
PARAMS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_output_mat', 'wildcard_mat', 'h0', 'hT', 'beta_vec',
          'embedding.weight')


def load():
    with open(os.path.join(GOLDEN, 'decomp_train_small.json')) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, 'decomp_train_small.npz')), np.load(os.path.join(GOLDEN, 'decomp_small.npz'))


@pytest.mark.parametrize('k', range(10))
def test_train_oracle_matches_reference_loss_and_gradients(k):
    meta, g, base = load()
    cfg = meta['configs'][k]
    pre = 'c{}.'.format(k)
    p = {n: torch.from_numpy(g[pre + 'w.' + n]) for n in PARAMS}
    p['priority_mat'] = torch.from_numpy(g[pre + 'w.priority_mat'])
    if cfg.get('use_crf'):
        p['crf.transitions'] = torch.from_numpy(g[pre + 'w.crf.transitions'])
    x, lengths, labels = torch.from_numpy(base['x']), torch.from_numpy(base['lengths']), torch.from_numpy(g['labels'])
    gate_names = tuple(n for n in GATES if pre + 'w.' + n in g.files)
    for n in gate_names:
        p[n] = torch.from_numpy(g[pre + 'w.' + n])
    loss, grads, _ = to.train_step(p, x, lengths, labels, nl=cfg['update_nonlinear'],
                                   additional_nonlinear=cfg.get('additional_nonlinear', 'none'),
                                   use_priority=bool(cfg.get('use_priority', 0)), farnn=cfg.get('farnn', 0),
                                   sig_k=float(cfg.get('sigmoid_exponent', 5)))
    assert abs(float(loss) - float(g[pre + 'loss'])) < 1e-5 * max(1.0, abs(float(g[pre + 'loss'])))
    for n in PARAMS + (('crf.transitions',) if cfg.get('use_crf') else ()) + gate_names:
        ref = g[pre + 'g.' + n]
        got = grads[n].numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6 * max(1.0, float(np.abs(ref).max())), err_msg=n)


@pytest.mark.parametrize('k', range(10))
def test_batched_train_oracle_equals_the_per_sequence_one(k):
    """The batch-vectorised form (used at full size and as the CPU baseline) against the per-sequence restatement."""
    meta, g, base = load()
    cfg = meta['configs'][k]
    pre = 'c{}.'.format(k)
    p = {n: torch.from_numpy(g[pre + 'w.' + n]) for n in PARAMS}
    p['priority_mat'] = torch.from_numpy(g[pre + 'w.priority_mat'])
    if cfg.get('use_crf'):
        p['crf.transitions'] = torch.from_numpy(g[pre + 'w.crf.transitions'])
    for n in GATES:
        if pre + 'w.' + n in g.files:
            p[n] = torch.from_numpy(g[pre + 'w.' + n])
    x, lengths, labels = torch.from_numpy(base['x']), torch.from_numpy(base['lengths']), torch.from_numpy(g['labels'])
    kw = dict(nl=cfg['update_nonlinear'], additional_nonlinear=cfg.get('additional_nonlinear', 'none'),
              use_priority=bool(cfg.get('use_priority', 0)), farnn=cfg.get('farnn', 0),
              sig_k=float(cfg.get('sigmoid_exponent', 5)))
    l1, g1, _ = to.train_step(p, x, lengths, labels, **kw)
    l2, g2, _ = to.train_step_batched(p, x, lengths, labels, **kw)
    assert abs(float(l1) - float(l2)) < 1e-5 * max(1.0, abs(float(l1)))
    assert set(g1) == set(g2)
    for n in g1:
        np.testing.assert_allclose(g2[n].numpy(), g1[n].numpy(), rtol=2e-4, atol=2e-6 * max(1.0, float(g1[n].abs().max())), err_msg=n)

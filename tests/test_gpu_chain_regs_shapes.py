"""Random geometries of the onehot i-FST through the one-launch tagging step (csrc/chain_regs.hip.h + beside.hip.h) against the
oracle: state counts 1..72 (every row-group / idle-lane layout of the register-fed recurrence) and 73..128 (its wide form,
csrc/chain_wide.hip.h: every ring width, with and without idle lanes, the reference's 104 states), label counts up to 256
(one to four decode columns per lane), sequence lengths up to 200 (several 16-token tiles per half, the 64-step block-address
window reloaded, the scorer wavefront scoring tiles alone while the chain runs), batches of 1..70, LOCAL and FULL mode, the four
non-linearities, both semirings, the priority matrix, scores asked for or not.  Bit-exact for integer-valued automata
(none / relu), 1e-4 otherwise; tags and flat tags equal the oracle's decode of the oracle's scores wherever the decision margin
exceeds twice the score bar.

    FARNN_SHAPE_SOAK=<n> raises the number of random configurations (default 40 in the suite), FARNN_SHAPE_SEED=<s> draws others.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import farnn_oracle as fo                    # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from util import NO_SWITCH                               # noqa: E402

pytestmark = pytest.mark.gpu


def _one(rng, it):
    from re2nn_seq_amd import _lib, synth
    S = int(rng.choice([1, 2, 5, 15, 16, 17, 31, 48, 63, 64, 65, 68, 71, 72] if rng.rand() < 0.5 else
                       [73, 76, 84, 85, 95, 96, 97, 104, 108, 109, 113, 120, 121, 125, 127, 128]))
    C = int(rng.choice([2, 9, 63, 64, 65, 128, 129, 200, 256]))
    L = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 65, 100, 129, 200]))
    B = int(rng.choice([1, 2, 7, 33, 70, 140, 300]))
    nl = str(rng.choice(['none', 'relu', 'tanh', 'relutanh']))
    semiring = str(rng.choice(['sum', 'sum', 'max']))
    full = bool(rng.rand() < 0.4)
    use_P = bool(rng.rand() < 0.25)
    want_scores = bool(rng.rand() < 0.6)
    V = 37
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=max(2.0, S / 5), n_final=min(2, S))
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    if rng.rand() < 0.3:
        lengths[:] = L
    sem = fo.SEMIRING_MAX if semiring == 'max' else fo.SEMIRING_SUM
    nlcodes = {'none': fo.NL_NONE, 'relu': fo.NL_RELU, 'tanh': fo.NL_TANH, 'relutanh': fo.NL_RELUTANH}
    if nl in ('none', 'relu'):
        with np.errstate(all='ignore'):
            ref = fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths, nl=nlcodes[nl], semiring=sem, P=None)
        if not np.isfinite(ref).all() or float(np.abs(ref).max()) >= 2.0 ** 22:
            nl = 'tanh'                                   # path counts of a tiny automaton over a long sentence leave fp32's exact range
    if nl in ('tanh', 'relutanh'):
        T = (T * 0.6).astype(np.float32)                  # keep the states bounded and non-trivial
    P = None
    if use_P:
        P = np.eye(C, dtype=np.float32) + (rng.rand(C, C) < 0.05).astype(np.float32) * 0.5
    # (round 6: the label-map path's default is two launches at every batch size; every second draw that takes it asks for the
    #  one-launch form instead -- FARNN_FUSE=1, read when the handle is created -- so that both forms stay under the random shapes)
    force_fuse = NO_SWITCH and P is None and not want_scores and S <= 72 and bool(it & 1)
    if force_fuse:
        os.environ['FARNN_FUSE'] = '1'
    try:
        h = _lib.create_onehot_ifst(T, W, O, h0, hT, P=P, nl=nl, semiring=semiring, threshold=0.5, o_idx=1 % C)
    finally:
        if force_fuse:
            del os.environ['FARNN_FUSE']
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
    mode = _lib.MODE_FULL if full else _lib.MODE_LOCAL
    tags = torch.full((B, L), -7, dtype=torch.int32, device='cuda')
    flat = torch.full((int(lengths.sum()),), -7, dtype=torch.int64, device='cuda')
    scores = torch.full((B, L, C), -7.0, dtype=torch.float32, device='cuda') if want_scores else None
    h.tag(xd.data_ptr(), ld.data_ptr(), B, L, mode, tags.data_ptr(), flat.data_ptr(),
          scores.data_ptr() if want_scores else None)
    torch.cuda.synchronize()
    name = h.kernel_name(_lib.KERN_CHAIN)
    h.close()
    nlc = nlcodes[nl]
    ref = fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths, nl=nlc, semiring=sem, P=P)
    exact = nl in ('none', 'relu') and not use_P
    mask = np.arange(L)[None, :] < lengths[:, None]
    live = np.ones_like(mask) if full else mask
    what = 'S={} C={} L={} B={} nl={} {} full={} P={} scores={} [{}]'.format(S, C, L, B, nl, semiring, full, use_P, want_scores, name)
    label_map = P is None and not want_scores              # (every generated output matrix is a label map)
    if NO_SWITCH:                                          # (which kernel ran: the default dispatch only; results: under every switch)
        assert name.startswith('chain_regs_kernel' if S <= 72 else 'chain_wide_kernel'), what      # the register-fed recurrence ...
        if L <= 64 and C <= 128 and (S > 72 or not label_map or force_fuse):
            assert 'fused' in name, what                   # ... in its one-launch form while states + score tiles fit half a CU's LDS
        if S <= 72 and label_map and not force_fuse:
            assert 'fused' not in name, what               # (rounds 5-6: the recurrence kernel + the label-map score launch: the faster form)
    tol = 0.0 if exact else 1e-4 * max(1.0, float(np.abs(ref).max()))
    if want_scores:
        got = scores.cpu().numpy()
        assert np.isfinite(got).all(), what
        assert float(np.abs(got[live] - ref[live]).max()) <= tol, what
        if not full:
            assert (got[~mask] == 0).all(), what
    rt = fo.decode_argmax(ref, 0.5, 1 % C)
    refc = ref.copy(); refc[..., -1] = np.minimum(refc[..., -1], 0.5)
    top2 = np.sort(refc, axis=-1)[..., -2:] if C > 1 else None
    safe = np.ones_like(mask) if exact else (top2[..., 1] - top2[..., 0]) > 2e-4
    tg = tags.cpu().numpy()
    assert np.array_equal(tg[live & safe], rt[live & safe]), what
    if not full:
        assert (tg[~mask] == -1).all(), what
    fl = flat.cpu().numpy()
    want_flat = fo.forward_local_tags(ref, lengths, 0.5, 1 % C)
    fsafe = safe[mask]
    assert np.array_equal(fl[fsafe], want_flat[fsafe]), what
    return what


def test_chain_regs_random_shapes_vs_oracle():
    n = int(os.environ.get('FARNN_SHAPE_SOAK', '60'))
    rng = np.random.RandomState(int(os.environ.get('FARNN_SHAPE_SEED', '20261003')))
    seen, fused = set(), 0
    for it in range(n):
        what = _one(rng, it)
        fused += 'fused' in what
        seen.add(what.split(' [')[0].split(' nl=')[0])
    assert len(seen) >= min(n, 25)                         # the draw really covered many geometries
    assert not NO_SWITCH or fused * 3 >= n                 # and a good share of them in the one-launch form

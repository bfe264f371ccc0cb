"""BASELINE.json configs[4] at FULL size on one GPU (its per-GPU shard: V=20 000, S=512, C=256,
L=128, B=1024 -- 21 GB of fp32 transition blocks, 44 GB touched per pass).  The oracle cannot run
that in seconds, so parity at this size goes through size-independent properties plus an exact
oracle check of a few whole sequences (their transition blocks are gathered back from the device,
so the oracle sees exactly the weights the kernel saw)."""
import numpy as np
import pytest
import torch

from oracle import farnn_oracle as fo

pytestmark = pytest.mark.gpu

V, S, C, B, L = 20000, 512, 256, 1024, 128
O_IDX = 3


@pytest.fixture(scope='module')
def big():
    from re2nn_seq_amd import _lib, synth
    if torch.cuda.mem_get_info()[0] < 100e9:
        pytest.skip('needs ~70 GB of free HBM')
    dv = torch.device('cuda', 0)
    g = torch.Generator(device='cuda'); g.manual_seed(99)
    # automaton-like 0/1 tensor, ~1 successor per (word, from-state): path counts stay tiny integers,
    # so fp32 sums are exact and the comparison with the oracle is bit-for-bit
    T = torch.empty((V, S, S), dtype=torch.float32, device=dv)
    for v0 in range(0, V, 500):
        T[v0:v0 + 500] = (torch.rand((min(500, V - v0), S, S), device=dv, generator=g) < 1.0 / S).float()
    T[V - 1] = 0
    W = torch.zeros((S, S), device=dv); W[0, 0] = 1; W[S - 1, S - 1] = 1
    O = torch.zeros((C, S), device=dv)
    O[torch.randint(0, C - 1, (S,), device=dv, generator=g), torch.arange(S, device=dv)] = 1
    O[:, 0] = 0; O[:, S - 1] = 0; O[C - 1, 0] = 1; O[C - 1, S - 1] = 1
    h0 = torch.zeros(S, device=dv); h0[0] = 1
    hT = torch.ones(S, device=dv)
    h = _lib.create_onehot_ifst(T, W, O, h0, hT, nl='none', o_idx=O_IDX, device=0)
    rng = np.random.RandomState(7)
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)
    lengths[0], lengths[1] = L, 1
    x[0] = rng.randint(0, V - 1, size=L); x[1, 1:] = V - 1
    yield dict(h=h, T=T, W=W.cpu().numpy(), O=O.cpu().numpy(), h0=h0.cpu().numpy(), hT=hT.cpu().numpy(),
               x=x, lengths=lengths, dv=dv)
    h.close()


def _tag(big, x, lengths, mode, want_scores=False):
    from re2nn_seq_amd import _lib
    b, l = x.shape
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(big['dv'])
    ld = torch.from_numpy(np.ascontiguousarray(lengths)).to(big['dv'])
    tags = torch.full((b, l), -7, dtype=torch.int32, device=big['dv'])
    flat = torch.full((int(lengths.sum()),), -7, dtype=torch.int64, device=big['dv'])
    scores = torch.empty((b, l, C), dtype=torch.float32, device=big['dv']) if want_scores else None
    big['h'].tag(xd.data_ptr(), ld.data_ptr(), b, l, mode, tags.data_ptr(),
                 flat.data_ptr() if mode == _lib.MODE_LOCAL else None,
                 scores.data_ptr() if want_scores else None)
    torch.cuda.synchronize()
    return tags.cpu().numpy(), flat.cpu().numpy(), None if scores is None else scores.cpu().numpy()


def test_fullsize_local_full_flat_consistency(big):
    from re2nn_seq_amd import _lib
    x, lengths = big['x'], big['lengths']
    tl, flat, _ = _tag(big, x, lengths, _lib.MODE_LOCAL)
    tf, _, _ = _tag(big, x, lengths, _lib.MODE_FULL)
    mask = np.arange(L)[None, :] < lengths[:, None]
    assert (tl[~mask] == -1).all()                          # LOCAL marks pads
    assert np.array_equal(tl[mask], tf[mask])               # pad steps of FULL mode do not leak back
    assert (tf >= 0).all() and (tf < C).all()
    assert np.array_equal(flat, tl[mask].astype(np.int64))  # batch-major flatten (utils.py:153-164)
    assert len(np.unique(flat)) > 20                        # the automaton does emit labels
    big['tags_local'] = tl


def test_fullsize_batch_permutation_and_idempotence(big):
    """Sequences are independent: any row permutation (which changes the internal length-sorted
    launch order and which sequences share a CU) permutes the output rows and nothing else; an
    interleaved call with another geometry on the same handle leaves no state behind."""
    from re2nn_seq_amd import _lib
    x, lengths = big['x'], big['lengths']
    ref = big.get('tags_local')
    if ref is None:
        ref = _tag(big, x, lengths, _lib.MODE_LOCAL)[0]
    perm = np.random.RandomState(3).permutation(B)
    tp, _, _ = _tag(big, x[perm], lengths[perm], _lib.MODE_LOCAL)
    assert np.array_equal(tp, ref[perm])
    _tag(big, x[:7, :5].copy(), np.minimum(lengths[:7], 5), _lib.MODE_LOCAL)
    again, _, _ = _tag(big, x, lengths, _lib.MODE_LOCAL)
    assert np.array_equal(again, ref)
    # a sub-batch gives the same rows as the full batch
    sub, _, _ = _tag(big, x[100:164].copy(), lengths[100:164].copy(), _lib.MODE_LOCAL)
    assert np.array_equal(sub, ref[100:164])


def test_fullsize_whole_sequences_vs_oracle(big):
    """Longest, shortest and four random sequences, bit-exact against the numpy oracle."""
    from re2nn_seq_amd import _lib
    x, lengths = big['x'], big['lengths']
    rows = np.array([0, 1, 17, 333, 700, B - 1])
    xs, ls = x[rows].copy(), lengths[rows].copy()
    toks, inv = np.unique(xs, return_inverse=True)
    Tsub = big['T'][torch.from_numpy(toks).to(big['dv'])].cpu().numpy()     # the blocks these rows touch
    xc = inv.reshape(xs.shape).astype(np.int64)
    ref = fo.onehot_ifst_scores(Tsub, big['W'], big['O'], big['h0'], big['hT'], xc, ls)
    _, _, scores = _tag(big, x, lengths, _lib.MODE_LOCAL, want_scores=True)
    Lmax = int(ls.max())
    mask = np.arange(Lmax)[None, :] < ls[:, None]
    got = scores[rows][:, :Lmax]
    assert np.array_equal(got[mask], ref[mask])
    assert ref[mask].max() >= 1.0                           # non-trivial: accepting paths exist
    tags = _tag(big, x, lengths, _lib.MODE_LOCAL)[0][rows][:, :Lmax]
    assert np.array_equal(tags[mask].astype(np.int64), fo.decode_argmax(ref, 0.5, O_IDX)[mask])


def test_fullsize_compact_form_equals_the_dense_blocks(big):
    """SURVEY.md 8f2 at BASELINE's largest config: the bit-packed blocks (0.66 GB beside 2 x 21 GB of fp32) with the
    active-state walk give the same scores and tags as the dense kernel, bit for bit (integer-valued states), in LOCAL
    and FULL mode; the time of both is printed for the record."""
    import time
    from re2nn_seq_amd import _lib
    h = big['h']
    assert h.has_compact()
    x, lengths = big['x'], big['lengths']
    dense = {}
    for mode in (_lib.MODE_LOCAL, _lib.MODE_FULL):
        dense[mode] = _tag(big, x, lengths, mode, want_scores=True)
    t0 = time.perf_counter(); _tag(big, x, lengths, _lib.MODE_LOCAL); t_dense = time.perf_counter() - t0
    h.set_compact(True)
    try:
        assert h.kernel_name(_lib.KERN_CHAIN) == 'compact_chain_kernel'
        for mode in (_lib.MODE_LOCAL, _lib.MODE_FULL):
            tags, flat, scores = _tag(big, x, lengths, mode, want_scores=True)
            assert np.array_equal(scores, dense[mode][2])
            assert np.array_equal(tags, dense[mode][0]) and np.array_equal(flat, dense[mode][1])
        t0 = time.perf_counter(); _tag(big, x, lengths, _lib.MODE_LOCAL); t_compact = time.perf_counter() - t0
        print('config 5 shard, one batch incl. copies: dense {:.1f} ms, compact {:.1f} ms'.format(t_dense * 1e3, t_compact * 1e3))
    finally:
        h.set_compact(False)

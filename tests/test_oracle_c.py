"""Pin the C port of the oracle (oracle/farnn_oracle.c, used as bench.py's cpu_baseline) to the
same reference fixtures as the numpy oracle."""
import numpy as np
import pytest

from oracle import c_port
from oracle import farnn_oracle as fo
from util import load_golden

NL = {'none': 0, 'relu': 1, 'tanh': 2, 'relutanh': 3}


@pytest.mark.parametrize('nl', ['none', 'relu', 'tanh', 'relutanh'])
@pytest.mark.parametrize('mode', ['sum', 'max'])
def test_c_port_ifst_small(nl, mode):
    g = load_golden('ifst_small')
    x, l, o_idx = g['x'], g['lengths'], int(g['o_idx'])
    Tf = (g['T'].astype(np.float32) + g['W'].astype(np.float32))
    tags, scores, _ = c_port.onehot_ifst_tag(Tf, g['O'], g['h0'], g['hT'], x, l, NL[nl],
                                             1 if mode == 'max' else 0, 0.5, o_idx, want_scores=True,
                                             nthreads=2)
    key = '{}.{}.p0.'.format(nl, mode)
    ref = g[key + 'scores']
    for b in range(x.shape[0]):
        n = int(l[b])
        if nl in ('none', 'relu'):
            assert np.array_equal(scores[b, :n], ref[b, :n])
        else:
            np.testing.assert_allclose(scores[b, :n], ref[b, :n], rtol=1e-4, atol=1e-4)
        assert (tags[b, n:] == -1).all()
    assert np.array_equal(fo.flatten(tags, l).astype(np.int64), g[key + 'flat_pred'])


def test_c_port_atis_scale():
    from re2nn_seq_amd import synth
    g = load_golden('atis_ifst')
    V, S, C, B, L = [int(v) for v in g['dims']]
    rng = np.random.RandomState(int(g['seed']))
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng)
    x = g['x'].astype(np.int64); l = g['lengths'].astype(np.int64)
    tags, _, used = c_port.onehot_ifst_tag(T + W, O, h0, hT, x, l)
    assert used >= 1
    assert np.array_equal(fo.flatten(tags, l).astype(np.int64), g['flat_pred'].astype(np.int64))


@pytest.mark.parametrize('reps,nthreads', [(1, 1), (3, 4), (5, 8)])
def test_c_port_throughput_form_gives_the_same_tags_and_scores(reps, nthreads):
    """bench.py's cpu_baseline on all host cores (round 5): the barrier-free form -- reps x B whole sequences dealt to the
    threads -- against the reference's own ATIS-scale outputs and bit-equal to the two-phase form."""
    from re2nn_seq_amd import synth
    g = load_golden('atis_ifst')
    V, S, C, B, L = [int(v) for v in g['dims']]
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, np.random.RandomState(int(g['seed'])))
    x = g['x'].astype(np.int64); l = g['lengths'].astype(np.int64)
    tags, scores, used = c_port.onehot_ifst_tag(T + W, O, h0, hT, x, l, want_scores=True, nthreads=nthreads, reps=reps, stream=True)
    assert used == nthreads
    assert np.array_equal(fo.flatten(tags, l).astype(np.int64), g['flat_pred'].astype(np.int64))
    t2, s2, _ = c_port.onehot_ifst_tag(T + W, O, h0, hT, x, l, want_scores=True, nthreads=2)
    assert np.array_equal(tags, t2) and np.array_equal(scores, s2)

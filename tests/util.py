"""Helpers shared by the tests."""
import argparse
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def ns(**kw):
    """The fields the model classes read from the reference's `args` Namespace (main.py:14-100)."""
    d = dict(rand_constant=0.0, train_wildcard=0, train_wildcard_wildcard=0, margin=0.3,
             threshold=0.5, train_mode='sum', local_loss_func='CE1', use_priority=0,
             independent=2, update_nonlinear='none', additional_states=0, train_word_embed=0,
             use_crf=0, random=0, train_h0=0, train_hT=0, train_V_embed=0, train_c_output=1,
             farnn=0, xavier=0, bias_init=5.0, sigmoid_exponent=5, beta=1.0, train_beta=0,
             additional_nonlinear='none', random_pad_func='uniform', marryup_type='none',
             c1_kdpr=1.0, c2_kdpr=1.0, c3_pr=1.0)
    d.update(kw)
    return argparse.Namespace(**d)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def assert_scores(sc, ref, exact):
    if exact:
        assert np.array_equal(sc, ref), 'max abs diff {}'.format(np.abs(sc - ref).max())
    else:
        np.testing.assert_allclose(sc, ref, rtol=1e-4, atol=1e-4)

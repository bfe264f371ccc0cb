"""Helpers shared by the tests."""
import argparse
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# FARNN_* switches the suite was STARTED under (scripts/gpu_r05_switches.sh runs it once per supported switch): a test's assertions
# about WHICH kernel ran hold for the default dispatch only; everything about results holds under every switch.
EXTERNAL_SWITCHES = sorted(k for k in os.environ if k.startswith('FARNN_') and k not in ('FARNN_LIB', 'FARNN_RCCL_LIB', 'FARNN_AB_CHILD') and
                           not k.startswith(('FARNN_SOAK', 'FARNN_SHAPE', 'FARNN_D1_SOAK', 'FARNN_BENCH')))
NO_SWITCH = not EXTERNAL_SWITCHES
AB_ONLY_SWITCHES = ('FARNN_CV_ONE', 'FARNN_CV_STASH', 'FARNN_NODEST')      # forms compiled into the A/B build only (csrc/build.py --probes)


def ab_build():
    """the loaded library carries the A/B-only forms (farnn_ab_build)"""
    from re2nn_seq_amd import _lib
    return _lib.ab_build()


def run_module_in_ab_build(path, extra_env=None, k=None, timeout=1500):
    """Runs a test module again in a child pytest with FARNN_LIB = the A/B build, where its A/B-only cases are not skipped.
    Returns None when there is nothing to do (already the A/B build, or it was not built)."""
    import subprocess
    import sys
    from re2nn_seq_amd import _lib
    if _lib.ab_build() or os.environ.get('FARNN_AB_CHILD') or not os.path.exists(_lib.AB_LIB_PATH):
        return None
    env = dict(os.environ, FARNN_LIB=_lib.AB_LIB_PATH, FARNN_AB_CHILD='1')
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'pytest', path, '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + (['-k', k] if k else [])
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


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


def in_float64(fn, params, *args, **kw):
    """`fn(params, ...)` of the oracle evaluated in float64 (fo.precision): every float array of `params` widened first."""
    from oracle import farnn_oracle as fo
    wide = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == 'f' else v) for k, v in params.items()}
    kw = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == 'f' else v) for k, v in kw.items()}
    with fo.precision(np.float64):
        return fn(wide, *args, **kw)


def assert_float_path(got, ref32, ref64, tol=1e-4, err_msg=''):
    """ONE rule for a floating-point kernel (north_star: "within 1e-4"): within `tol` (absolute + relative) of the EXACT value
    -- the oracle evaluated in float64 -- and never farther from the float32 oracle than `tol` plus that oracle's own distance
    from the exact value at the same entry (two float32 evaluations of a sensitive recurrence scatter around the exact value;
    neither is the other's yardstick beyond its own noise).  No per-shape bars."""
    got, ref32, ref64 = np.asarray(got, np.float64), np.asarray(ref32, np.float64), np.asarray(ref64, np.float64)
    e64 = np.abs(got - ref64)
    bad = e64 > tol * (1.0 + np.abs(ref64))
    assert not bad.any(), '{} {} of {} entries beyond {} of the float64 value; worst {:.3e}'.format(
        err_msg, int(bad.sum()), bad.size, tol, float((e64 / (1.0 + np.abs(ref64))).max()))
    e32 = np.abs(got - ref32)
    bad = e32 > tol * (1.0 + np.abs(ref32)) + np.abs(ref32 - ref64)
    assert not bad.any(), '{} {} of {} entries beyond {} + the float32 oracle\'s own error; worst {:.3e}'.format(
        err_msg, int(bad.sum()), bad.size, tol, float(e32.max()))

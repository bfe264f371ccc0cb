"""ctypes wrapper of oracle/libfarnn_oracle.so (the C port of the oracle; test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libfarnn_oracle.so')
_lib = None


def build(native=False):
    """native=True rebuilds with -march=native for the machine this runs on (bench cpu_baseline)."""
    if native:
        subprocess.run(['make', '-B', '-C', _HERE, 'native'], check=True, stdout=subprocess.DEVNULL)
        return os.path.join(_HERE, 'libfarnn_oracle_native.so')
    subprocess.run(['make', '-C', _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


def load(native=False):
    global _lib
    if native:
        _lib = None
    if _lib is None:
        so = build(native=True) if native else _SO
        if not os.path.exists(so):
            build()
        lib = C.CDLL(so)
        lib.oracle_onehot_ifst_tag.restype = C.c_int
        lib.oracle_onehot_ifst_tag.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 2 + \
            [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.oracle_onehot_ifst_tag_reps.restype = C.c_int
        lib.oracle_onehot_ifst_tag_reps.argtypes = lib.oracle_onehot_ifst_tag.argtypes + [C.c_int]
        lib.oracle_onehot_ifst_tag_stream.restype = C.c_int
        lib.oracle_onehot_ifst_tag_stream.argtypes = lib.oracle_onehot_ifst_tag_reps.argtypes
        lib.oracle_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def onehot_ifst_tag(Tf, O, h0, hT, x, lengths, nl=0, semiring=0, threshold=0.5, o_idx=0,
                    want_scores=False, nthreads=0, reps=1, stream=False):
    """Tf = T + W premixed (float32 [V,S,S]).  Returns (tags int32 [B,L], scores or None, threads).
    reps > 1: the same batch that many times inside one parallel region (bench.py's cpu_baseline: a hot thread team).
    stream=True: the barrier-free throughput form (reps x B whole sequences dealt to the threads; same tags)."""
    lib = load()
    Tf = np.ascontiguousarray(Tf, np.float32); O = np.ascontiguousarray(O, np.float32)
    h0 = np.ascontiguousarray(h0, np.float32); hT = np.ascontiguousarray(hT, np.float32)
    x = np.ascontiguousarray(x, np.int64); lengths = np.ascontiguousarray(lengths, np.int64)
    V, S, _ = Tf.shape
    Cn = O.shape[0]
    B, L = x.shape
    tags = np.empty((B, L), np.int32)
    scores = np.empty((B, L, Cn), np.float32) if want_scores else None
    used = (lib.oracle_onehot_ifst_tag_stream if stream else lib.oracle_onehot_ifst_tag_reps)(
        Tf.ctypes.data, O.ctypes.data, h0.ctypes.data, hT.ctypes.data, V, S, Cn, x.ctypes.data,
        lengths.ctypes.data, B, L, int(nl), int(semiring), float(threshold), int(o_idx),
        tags.ctypes.data, scores.ctypes.data if want_scores else None, int(nthreads), int(reps))
    return tags, scores, used

"""ctypes binding of the C-ABI in include/farnn.h (libfarnn_hip.so).

There is no CPU fallback: if the shared library has not been built (``__graft_entry__.build()``
or ``python -m re2nn_seq_amd.csrc.build``) or no MI355X is visible, the functions here raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FARNN_LIB: a diagnostic build of the same library (csrc/build.py --probes); never a different implementation
LIB_PATH = os.environ.get('FARNN_LIB') or os.path.join(_HERE, 'csrc', 'libfarnn_hip.so')

OK = 0
NL = {'none': 0, 'relu': 1, 'tanh': 2, 'relutanh': 3, 'sigmoid': 4}
SEMIRING = {'sum': 0, 'max': 1}
MODE_LOCAL, MODE_FULL, MODE_RE = 0, 1, 2
HOST_SLOTS = 4                  # FARNN_HOST_SLOTS: batches the host-buffer path keeps in flight
KERN_CHAIN, KERN_SCORE, KERN_PREP = 0, 1, 2

_f32p = C.POINTER(C.c_float)


class FarnnError(RuntimeError):
    pass


class OnehotIfstDesc(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('C', C.c_int32),
                ('T', _f32p), ('W', _f32p), ('O', _f32p), ('h0', _f32p), ('hT', _f32p), ('P', _f32p),
                ('nl', C.c_int32), ('semiring', C.c_int32), ('threshold', C.c_float),
                ('o_idx', C.c_int32), ('use_crf', C.c_int32), ('crf_trans', _f32p),
                ('weights_on_device', C.c_int32)]


class OnehotFst4Desc(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('C', C.c_int32),
                ('T4', _f32p), ('W4', _f32p), ('h0', _f32p), ('hT', _f32p), ('P', _f32p),
                ('semiring', C.c_int32), ('threshold', C.c_float), ('o_idx', C.c_int32),
                ('weights_on_device', C.c_int32)]


class OnehotInd1Desc(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('C', C.c_int32),
                ('T', _f32p), ('W', _f32p), ('Oten', _f32p), ('h0', _f32p), ('hT', _f32p), ('P', _f32p),
                ('semiring', C.c_int32), ('mask_by_output', C.c_int32), ('threshold', C.c_float),
                ('o_idx', C.c_int32), ('weights_on_device', C.c_int32)]


class DecompIfstDesc(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('R', C.c_int32), ('K', C.c_int32),
                ('Vgen', _f32p), ('S1', _f32p), ('S2', _f32p), ('W', _f32p), ('Cout', _f32p),
                ('h0', _f32p), ('hT', _f32p), ('P', _f32p),
                ('farnn', C.c_int32),
                ('Wss1', _f32p), ('Wrs1', _f32p), ('bs1', _f32p),
                ('Wss2', _f32p), ('Wrs2', _f32p), ('bs2', _f32p),
                ('sigmoid_exponent', C.c_float), ('nl', C.c_int32), ('semiring', C.c_int32),
                ('threshold', C.c_float), ('o_idx', C.c_int32), ('use_crf', C.c_int32),
                ('crf_trans', _f32p), ('weights_on_device', C.c_int32)]


class VgenFold(C.Structure):
    _fields_ = [('V_embed', _f32p), ('E', _f32p), ('G', _f32p), ('beta', _f32p), ('D', C.c_int32), ('add_nl', C.c_int32),
                ('normalize', C.c_int32), ('on_device', C.c_int32)]


NORM = {'none': 0, 'l1': 1, 'l2': 2, 'l1-rank': 3, 'l2-rank': 4}


class DecompInd1Desc(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('R', C.c_int32), ('RO', C.c_int32), ('K', C.c_int32),
                ('Vgen', _f32p), ('S1', _f32p), ('S2', _f32p), ('W', _f32p), ('Cout', _f32p),
                ('S1o', _f32p), ('S2o', _f32p), ('Wo', _f32p),
                ('h0', _f32p), ('hT', _f32p), ('P', _f32p),
                ('farnn', C.c_int32),
                ('Wss1', _f32p), ('Wrs1', _f32p), ('bs1', _f32p),
                ('Wss2', _f32p), ('Wrs2', _f32p), ('bs2', _f32p),
                ('sigmoid_exponent', C.c_float), ('nl', C.c_int32), ('semiring', C.c_int32),
                ('threshold', C.c_float), ('o_idx', C.c_int32), ('use_crf', C.c_int32),
                ('crf_trans', _f32p), ('weights_on_device', C.c_int32)]


class EdgeList(C.Structure):
    _fields_ = [('n_edges', C.c_int64),
                ('word', C.POINTER(C.c_int32)), ('from_', C.POINTER(C.c_int32)), ('to', C.POINTER(C.c_int32)),
                ('label', C.POINTER(C.c_int32)), ('val', _f32p)]


class DecompFstDesc(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('R', C.c_int32), ('RW', C.c_int32), ('K', C.c_int32),
                ('Vgen', _f32p), ('C', _f32p), ('S1', _f32p), ('S2', _f32p), ('Cw', _f32p),
                ('S1w', _f32p), ('S2w', _f32p), ('WW', _f32p),
                ('h0', _f32p), ('hT', _f32p), ('P', _f32p),
                ('farnn', C.c_int32),
                ('Wss1', _f32p), ('Wrs1', _f32p), ('bs1', _f32p),
                ('Wss2', _f32p), ('Wrs2', _f32p), ('bs2', _f32p),
                ('sigmoid_exponent', C.c_float), ('nl', C.c_int32), ('semiring', C.c_int32),
                ('threshold', C.c_float), ('o_idx', C.c_int32), ('use_crf', C.c_int32),
                ('crf_trans', _f32p), ('weights_on_device', C.c_int32)]


# every symbol include/farnn.h declares, with its ctypes signature (tests check the exports)
_vp = C.c_void_p
class TrainDims(C.Structure):
    _fields_ = [('V', C.c_int32), ('S', C.c_int32), ('R', C.c_int32), ('K', C.c_int32), ('nl', C.c_int32),
                ('threshold', C.c_float), ('o_idx', C.c_int32), ('farnn', C.c_int32), ('sigmoid_exponent', C.c_float),
                ('use_crf', C.c_int32)]


class TrainWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('Vgen', 'S1', 'S2', 'W', 'C', 'h0', 'hT', 'P', 'crf_trans',
                                          'Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')]


class TrainOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('loss', 'dVgen', 'dS1', 'dS2', 'dW', 'dC', 'dh0', 'dhT', 'tags', 'dtrans',
                                          'dWss1', 'dWrs1', 'dbs1', 'dWss2', 'dWrs2', 'dbs2')]


SIGNATURES = {
    'farnn_train_create': (C.c_int, [C.POINTER(TrainDims), C.c_int, C.POINTER(C.c_void_p)]),
    'farnn_train_destroy': (None, [C.c_void_p]),
    'farnn_decomp_ifst_train_step': (C.c_int, [C.c_void_p, C.POINTER(TrainWeights), C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                               C.POINTER(TrainOutputs), C.c_void_p]),
    'farnn_train_set_profiling': (C.c_int, [C.c_void_p, C.c_int32]),
    'farnn_train_time': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'farnn_onehot_ifst_create': (C.c_int, [C.POINTER(OnehotIfstDesc), C.c_int, C.POINTER(_vp)]),
    'farnn_onehot_ifst_create_from_edges': (C.c_int, [C.POINTER(OnehotIfstDesc), C.POINTER(EdgeList), C.c_int,
                                                      C.POINTER(_vp)]),
    'farnn_onehot_ifst_create_compact': (C.c_int, [C.POINTER(OnehotIfstDesc), C.POINTER(EdgeList), C.c_int,
                                                   C.POINTER(_vp)]),
    'farnn_has_compact': (C.c_int, [_vp]),
    'farnn_set_compact': (C.c_int, [_vp, C.c_int32]),
    'farnn_onehot_fst4_create_from_edges': (C.c_int, [C.POINTER(OnehotFst4Desc), C.POINTER(EdgeList), C.c_int,
                                                      C.POINTER(_vp)]),
    'farnn_onehot_ind1_create_from_edges': (C.c_int, [C.POINTER(OnehotInd1Desc), C.POINTER(EdgeList), C.c_int,
                                                      C.POINTER(_vp)]),
    'farnn_onehot_fst4_create': (C.c_int, [C.POINTER(OnehotFst4Desc), C.c_int, C.POINTER(_vp)]),
    'farnn_onehot_ind1_create': (C.c_int, [C.POINTER(OnehotInd1Desc), C.c_int, C.POINTER(_vp)]),
    'farnn_decomp_ifst_create': (C.c_int, [C.POINTER(DecompIfstDesc), C.c_int, C.POINTER(_vp)]),
    'farnn_decomp_ifst_create_folded': (C.c_int, [C.POINTER(DecompIfstDesc), C.POINTER(VgenFold), C.c_int, C.POINTER(_vp)]),
    'farnn_decomp_ind1_create': (C.c_int, [C.POINTER(DecompInd1Desc), C.c_int, C.POINTER(_vp)]),
    'farnn_decomp_fst_create': (C.c_int, [C.POINTER(DecompFstDesc), C.c_int, C.POINTER(_vp)]),
    'farnn_tag': (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    'farnn_reserve': (C.c_int, [_vp, C.c_int32, C.c_int32]),
    'farnn_tag_host_submit': (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'farnn_flatten_host': (C.c_int64, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    'farnn_tag_host_wait': (C.c_int, [_vp, C.c_int32, _vp, C.POINTER(C.c_int64)]),
    'farnn_destroy': (None, [_vp]),
    'farnn_abi_version': (C.c_int, []),
    'farnn_ab_build': (C.c_int, []),
    'farnn_device_count': (C.c_int, []),
    'farnn_last_error': (C.c_char_p, []),
    'farnn_num_columns': (C.c_int, [_vp]),
    'farnn_algorithmic_bytes': (C.c_double, [_vp, C.c_int64]),
    'farnn_kernel_algorithmic_bytes': (C.c_double, [_vp, C.c_int32, C.c_int64]),
    'farnn_set_profiling': (C.c_int, [_vp, C.c_int32]),
    'farnn_kernel_time': (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'farnn_kernel_name': (C.c_char_p, [_vp, C.c_int32]),
}

_lib = None


def load():
    """Load libfarnn_hip.so (once).  Raises FarnnError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FarnnError(
            'HIP library not built: {} is missing. Run `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.'.format(LIB_PATH))
    # PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  The tensors this
    # library receives live in THAT runtime's address space, so it must be the one already loaded
    # when libfarnn_hip.so resolves its HIP symbols: import torch first.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


AB_LIB_PATH = os.path.join(_HERE, 'csrc', 'libfarnn_hip_probes.so')     # the A/B (profiling) build: csrc/build.py --probes


def ab_build():
    """True when the loaded library is the A/B build (it carries the forms the production build left behind: FARNN_CV_ONE)."""
    return bool(load().farnn_ab_build())


def check(rc, what=''):
    if rc != OK:
        msg = load().farnn_last_error()
        raise FarnnError('{} failed with code {}: {}'.format(what or 'farnn call', rc,
                                                             msg.decode() if msg else ''))


def flatten_host(a_ptr, len_ptr, B, L, out_ptr):
    """utils.flatten of a host int64 [B,L] array into `out` (farnn_flatten_host); returns the element count."""
    return load().farnn_flatten_host(a_ptr, len_ptr, B, L, out_ptr)


def f32(a):
    """float32 C-contiguous numpy view/copy (the reference stores float64 and calls .float())."""
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def ptr(a):
    return None if a is None else a.ctypes.data_as(_f32p)


class Handle:
    """Owns one farnn_model*."""

    def __init__(self, raw, keepalive=()):
        self._raw = raw
        self._keep = keepalive

    @property
    def raw(self):
        if not self._raw:
            raise FarnnError('model handle already destroyed')
        return self._raw

    def close(self):
        if self._raw:
            load().farnn_destroy(self._raw)
            self._raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thin wrappers -----------------------------------------------------------------------
    def num_columns(self):
        return load().farnn_num_columns(self.raw)

    def reserve(self, B, L):
        check(load().farnn_reserve(self.raw, B, L), 'farnn_reserve')

    def tag(self, x_ptr, len_ptr, B, L, mode, tags_ptr=None, flat_ptr=None, scores_ptr=None, stream=None):
        check(load().farnn_tag(self.raw, x_ptr, len_ptr, B, L, mode, tags_ptr, flat_ptr, scores_ptr,
                               stream), 'farnn_tag')

    def tag_host_submit(self, x_ptr, len_ptr, B, L):
        """Host buffers in (int64 [B,L], [B]); returns a ticket for tag_host_wait.  Work is enqueued, not awaited."""
        t, n = C.c_int32(-1), C.c_int64(0)
        check(load().farnn_tag_host_submit(self.raw, x_ptr, len_ptr, B, L, C.byref(t), C.byref(n)), 'farnn_tag_host_submit')
        return t.value, n.value

    def tag_host_wait(self, ticket, flat_ptr):
        n = C.c_int64(0)
        check(load().farnn_tag_host_wait(self.raw, ticket, flat_ptr, C.byref(n)), 'farnn_tag_host_wait')
        return n.value

    def has_compact(self):
        return bool(load().farnn_has_compact(self.raw))

    def set_compact(self, enable=True):
        """Bit-packed blocks + active-state walk instead of the dense fp32 blocks (0/1 automata; include/farnn.h)."""
        check(load().farnn_set_compact(self.raw, int(bool(enable))), 'farnn_set_compact')

    def algorithmic_bytes(self, valid_tokens):
        return load().farnn_algorithmic_bytes(self.raw, int(valid_tokens))

    def kernel_algorithmic_bytes(self, which, valid_tokens):
        return load().farnn_kernel_algorithmic_bytes(self.raw, int(which), int(valid_tokens))

    def set_profiling(self, every):
        """every=0 off; every=N: time every N-th farnn_tag call with HIP events."""
        check(load().farnn_set_profiling(self.raw, int(every)), 'farnn_set_profiling')

    def kernel_time(self, which):
        ms, n = C.c_double(0), C.c_int64(0)
        check(load().farnn_kernel_time(self.raw, which, C.byref(ms), C.byref(n)), 'farnn_kernel_time')
        return ms.value, n.value

    def kernel_name(self, which):
        return load().farnn_kernel_name(self.raw, which).decode()


def _create(fn_name, desc, device, keep):
    lib = load()
    out = _vp()
    check(getattr(lib, fn_name)(C.byref(desc), int(device), C.byref(out)), fn_name)
    return Handle(out, keep)


def _dev_ptr(t):
    return C.cast(C.c_void_p(t.data_ptr()), _f32p)


def create_onehot_ifst(T, W, O, h0, hT, P=None, nl='none', semiring='sum', threshold=0.5, o_idx=0,
                       use_crf=False, crf_trans=None, device=0):
    """Weights as numpy arrays (host, copied by the library) or -- all of them -- as float32
    torch tensors already on the target device (weights_on_device=1; nothing crosses PCIe)."""
    import torch
    if torch.is_tensor(T):
        arrs = [a for a in (T, W, O, h0, hT, P, crf_trans) if a is not None]
        assert all(torch.is_tensor(a) and a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()
                   for a in arrs), 'device weights must all be contiguous float32 CUDA tensors'
        V, S, _ = T.shape
        d = OnehotIfstDesc(V, S, O.shape[0], _dev_ptr(T), _dev_ptr(W), _dev_ptr(O), _dev_ptr(h0), _dev_ptr(hT),
                           None if P is None else _dev_ptr(P), NL[nl], SEMIRING[semiring], float(threshold),
                           int(o_idx), int(bool(use_crf)), None if crf_trans is None else _dev_ptr(crf_trans), 1)
        torch.cuda.synchronize(T.device)
        return _create('farnn_onehot_ifst_create', d, device, tuple(arrs))
    T, W, O, h0, hT = f32(T), f32(W), f32(O), f32(h0), f32(hT)
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    V, S, _ = T.shape
    d = OnehotIfstDesc(V, S, O.shape[0], ptr(T), ptr(W), ptr(O), ptr(h0), ptr(hT), ptr(P),
                       NL[nl], SEMIRING[semiring], float(threshold), int(o_idx), int(bool(use_crf)),
                       ptr(crf_trans), 0)
    return _create('farnn_onehot_ifst_create', d, device, (T, W, O, h0, hT, P, crf_trans))


def _edge_list(word, frm, to, label, val):
    i32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.int32)      # noqa: E731
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))                     # noqa: E731
    word, frm, to, label = i32(word), i32(frm), i32(to), i32(label)
    n = word.shape[0]
    assert frm.shape == to.shape == label.shape == (n,)
    val = None if val is None else f32(val)
    return EdgeList(n, ip(word), ip(frm), ip(to), ip(label), ptr(val)), (word, frm, to, label, val)


def _create_from_edges(fn_name, base, edges, device, keep):
    lib = load()
    out = _vp()
    check(getattr(lib, fn_name)(C.byref(base), C.byref(edges), int(device), C.byref(out)), fn_name)
    return Handle(out, keep)


def create_onehot_ifst_from_edges(V, S, n_cols, word, frm, to, label, h0, hT, val=None, P=None, nl='none',
                                  semiring='sum', threshold=0.5, o_idx=0, use_crf=False, crf_trans=None,
                                  device=0):
    """The i-FST built on the device from the automaton's edge list (farnn_edge_list): no dense
    [V,S,S] tensor exists on the host.  `word` = -1 marks a wildcard edge, < -1 an entry that only
    labels its destination state; `label` < 0: no label."""
    edges, keep = _edge_list(word, frm, to, label, val)
    h0, hT = f32(h0), f32(hT)
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    base = OnehotIfstDesc(int(V), int(S), int(n_cols), None, None, None, ptr(h0), ptr(hT), ptr(P),
                          NL[nl], SEMIRING[semiring], float(threshold), int(o_idx), int(bool(use_crf)),
                          ptr(crf_trans), 0)
    return _create_from_edges('farnn_onehot_ifst_create_from_edges', base, edges, device,
                              keep + (h0, hT, P, crf_trans))


def create_onehot_ifst_compact(V, S, n_cols, word, frm, to, label, h0, hT, val=None, P=None, nl='none', threshold=0.5,
                               o_idx=0, use_crf=False, crf_trans=None, device=0):
    """The i-FST in its compact form ONLY (bit-packed blocks scattered from the edge list; no dense tensor anywhere)."""
    edges, keep = _edge_list(word, frm, to, label, val)
    h0, hT = f32(h0), f32(hT)
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    base = OnehotIfstDesc(int(V), int(S), int(n_cols), None, None, None, ptr(h0), ptr(hT), ptr(P),
                          NL[nl], SEMIRING['sum'], float(threshold), int(o_idx), int(bool(use_crf)),
                          ptr(crf_trans), 0)
    return _create_from_edges('farnn_onehot_ifst_create_compact', base, edges, device, keep + (h0, hT, P, crf_trans))


def create_onehot_fst4_from_edges(V, S, n_cols, word, frm, to, label, h0, hT, val=None, P=None, semiring='sum',
                                  threshold=0.5, o_idx=0, device=0):
    """FARNN_S_O (4-D FST) from the edge list: T4[w,l,f,t] / W4[l,f,t] are scattered on the device."""
    edges, keep = _edge_list(word, frm, to, label, val)
    h0, hT = f32(h0), f32(hT)
    P = None if P is None else f32(P)
    base = OnehotFst4Desc(int(V), int(S), int(n_cols), None, None, ptr(h0), ptr(hT), ptr(P),
                          SEMIRING[semiring], float(threshold), int(o_idx), 0)
    return _create_from_edges('farnn_onehot_fst4_create_from_edges', base, edges, device, keep + (h0, hT, P))


def create_onehot_ind1_from_edges(V, S, n_cols, word, frm, to, label, h0, hT, val=None, P=None, semiring='sum',
                                  mask_by_output=False, threshold=0.5, o_idx=0, device=0):
    """FARNN_S_O_I (independent=1) from the edge list: T, W and Oten[l,f,t] are scattered on the device."""
    edges, keep = _edge_list(word, frm, to, label, val)
    h0, hT = f32(h0), f32(hT)
    P = None if P is None else f32(P)
    base = OnehotInd1Desc(int(V), int(S), int(n_cols), None, None, None, ptr(h0), ptr(hT), ptr(P),
                          SEMIRING[semiring], int(bool(mask_by_output)), float(threshold), int(o_idx), 0)
    return _create_from_edges('farnn_onehot_ind1_create_from_edges', base, edges, device, keep + (h0, hT, P))


def create_onehot_fst4(T4, W4, h0, hT, P=None, semiring='sum', threshold=0.5, o_idx=0, device=0):
    T4, W4, h0, hT = f32(T4), f32(W4), f32(h0), f32(hT)
    P = None if P is None else f32(P)
    V, Cn, S, _ = T4.shape
    d = OnehotFst4Desc(V, S, Cn, ptr(T4), ptr(W4), ptr(h0), ptr(hT), ptr(P), SEMIRING[semiring],
                       float(threshold), int(o_idx), 0)
    return _create('farnn_onehot_fst4_create', d, device, (T4, W4, h0, hT, P))


def create_onehot_ind1(T, W, Oten, h0, hT, P=None, semiring='sum', mask_by_output=False, threshold=0.5,
                       o_idx=0, device=0):
    T, W, Oten, h0, hT = f32(T), f32(W), f32(Oten), f32(h0), f32(hT)
    P = None if P is None else f32(P)
    V, S, _ = T.shape
    d = OnehotInd1Desc(V, S, Oten.shape[0], ptr(T), ptr(W), ptr(Oten), ptr(h0), ptr(hT), ptr(P),
                       SEMIRING[semiring], int(bool(mask_by_output)), float(threshold), int(o_idx), 0)
    return _create('farnn_onehot_ind1_create', d, device, (T, W, Oten, h0, hT, P))


def create_decomp_ifst(Vgen, S1, S2, W, Cout, h0, hT, P=None, farnn=0, gates=None, sigmoid_exponent=5,
                       nl='none', semiring='sum', threshold=0.5, o_idx=0, use_crf=False, crf_trans=None,
                       device=0):
    Vgen, S1, S2, W, Cout, h0, hT = (f32(a) for a in (Vgen, S1, S2, W, Cout, h0, hT))
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    g = {k: f32(v) for k, v in (gates or {}).items()}
    S, R = S1.shape
    d = DecompIfstDesc(Vgen.shape[0], S, R, Cout.shape[0], ptr(Vgen), ptr(S1), ptr(S2), ptr(W), ptr(Cout),
                       ptr(h0), ptr(hT), ptr(P), int(farnn),
                       ptr(g.get('Wss1')), ptr(g.get('Wrs1')), ptr(g.get('bs1')),
                       ptr(g.get('Wss2')), ptr(g.get('Wrs2')), ptr(g.get('bs2')),
                       float(sigmoid_exponent), NL[nl], SEMIRING[semiring], float(threshold), int(o_idx),
                       int(bool(use_crf)), ptr(crf_trans), 0)
    return _create('farnn_decomp_ifst_create', d, device, (Vgen, S1, S2, W, Cout, h0, hT, P, crf_trans, g))


def create_decomp_ifst_folded(V_embed, E, G, beta, S1, S2, W, Cout, h0, hT, add_nl='none', normalize='none', P=None,
                              farnn=0, gates=None, sigmoid_exponent=5, nl='none', semiring='sum', threshold=0.5, o_idx=0,
                              use_crf=False, crf_trans=None, device=0):
    """create_decomp_ifst with the word table (and the per-rank --normalize_automata scaling of V_embed, S1, S2) computed on
    the device: Vgen = V_embed * beta + nl_add(E @ G) * (1 - beta).  G is the bridge of the UN-normalised V_embed."""
    V_embed, E, G, beta, S1, S2, W, Cout, h0, hT = (f32(a) for a in (V_embed, E, G, beta, S1, S2, W, Cout, h0, hT))
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    g = {k: f32(v) for k, v in (gates or {}).items()}
    S, R = S1.shape
    d = DecompIfstDesc(V_embed.shape[0], S, R, Cout.shape[0], None, ptr(S1), ptr(S2), ptr(W), ptr(Cout),
                       ptr(h0), ptr(hT), ptr(P), int(farnn),
                       ptr(g.get('Wss1')), ptr(g.get('Wrs1')), ptr(g.get('bs1')),
                       ptr(g.get('Wss2')), ptr(g.get('Wrs2')), ptr(g.get('bs2')),
                       float(sigmoid_exponent), NL[nl], SEMIRING[semiring], float(threshold), int(o_idx),
                       int(bool(use_crf)), ptr(crf_trans), 0)
    fd = VgenFold(ptr(V_embed), ptr(E), ptr(G), ptr(beta), E.shape[1], NL[add_nl], NORM[normalize], 0)
    lib = load()
    out = _vp()
    check(lib.farnn_decomp_ifst_create_folded(C.byref(d), C.byref(fd), int(device), C.byref(out)),
          'farnn_decomp_ifst_create_folded')
    return Handle(out, (V_embed, E, G, beta, S1, S2, W, Cout, h0, hT, P, crf_trans, g))


def create_decomp_ind1(Vgen, S1, S2, W, Cout, S1o, S2o, h0, hT, Wo=None, P=None, farnn=0, gates=None,
                       sigmoid_exponent=5, nl='none', semiring='sum', threshold=0.5, o_idx=0, use_crf=False,
                       crf_trans=None, device=0):
    Vgen, S1, S2, W, Cout, S1o, S2o, h0, hT = (f32(a) for a in (Vgen, S1, S2, W, Cout, S1o, S2o, h0, hT))
    Wo = None if Wo is None else f32(Wo)
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    g = {k: f32(v) for k, v in (gates or {}).items()}
    S, R = S1.shape
    K, RO = Cout.shape
    d = DecompInd1Desc(Vgen.shape[0], S, R, RO, K, ptr(Vgen), ptr(S1), ptr(S2), ptr(W), ptr(Cout),
                       ptr(S1o), ptr(S2o), ptr(Wo), ptr(h0), ptr(hT), ptr(P), int(farnn),
                       ptr(g.get('Wss1')), ptr(g.get('Wrs1')), ptr(g.get('bs1')),
                       ptr(g.get('Wss2')), ptr(g.get('Wrs2')), ptr(g.get('bs2')),
                       float(sigmoid_exponent), NL[nl], SEMIRING[semiring], float(threshold), int(o_idx),
                       int(bool(use_crf)), ptr(crf_trans), 0)
    return _create('farnn_decomp_ind1_create', d, device,
                   (Vgen, S1, S2, W, Cout, S1o, S2o, Wo, h0, hT, P, crf_trans, g))


def create_decomp_fst(Vgen, Cemb, S1, S2, Cw, S1w, S2w, WW, h0, hT, P=None, farnn=0, gates=None,
                      sigmoid_exponent=5, nl='none', semiring='sum', threshold=0.5, o_idx=0, use_crf=False,
                      crf_trans=None, device=0):
    Vgen, Cemb, S1, S2, Cw, S1w, S2w, WW, h0, hT = (
        f32(a) for a in (Vgen, Cemb, S1, S2, Cw, S1w, S2w, WW, h0, hT))
    P = None if P is None else f32(P)
    crf_trans = None if crf_trans is None else f32(crf_trans)
    g = {k: f32(v) for k, v in (gates or {}).items()}
    S, R = S1.shape
    K, RW = Cw.shape
    if Cemb.shape != (K, R):
        raise FarnnError('C_embed must be [K,R] = {}, got {}'.format((K, R), Cemb.shape))
    d = DecompFstDesc(Vgen.shape[0], S, R, RW, K, ptr(Vgen), ptr(Cemb), ptr(S1), ptr(S2), ptr(Cw),
                      ptr(S1w), ptr(S2w), ptr(WW), ptr(h0), ptr(hT), ptr(P), int(farnn),
                      ptr(g.get('Wss1')), ptr(g.get('Wrs1')), ptr(g.get('bs1')),
                      ptr(g.get('Wss2')), ptr(g.get('Wrs2')), ptr(g.get('bs2')),
                      float(sigmoid_exponent), NL[nl], SEMIRING[semiring], float(threshold), int(o_idx),
                      int(bool(use_crf)), ptr(crf_trans), 0)
    return _create('farnn_decomp_fst_create', d, device,
                   (Vgen, Cemb, S1, S2, Cw, S1w, S2w, WW, h0, hT, P, crf_trans, g))


class TrainContext:
    """Owns one farnn_train_ctx* (training step of the decomposed i-FST, include/farnn.h)."""

    def __init__(self, V, S, R, K, nl='none', threshold=0.5, o_idx=0, device=0, use_crf=False, farnn=0,
                 sigmoid_exponent=5.0):
        d = TrainDims(int(V), int(S), int(R), int(K), NL[nl], float(threshold), int(o_idx), int(farnn),
                      float(sigmoid_exponent), int(bool(use_crf)))
        out = C.c_void_p()
        check(load().farnn_train_create(C.byref(d), int(device), C.byref(out)), 'farnn_train_create')
        self._raw = out
        self.dims = (int(V), int(S), int(R), int(K))

    def close(self):
        if self._raw:
            load().farnn_train_destroy(self._raw)
            self._raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, weights, x_ptr, len_ptr, labels_ptr, B, L, valid_tokens, outputs, stream=None):
        """weights / outputs: dicts of device pointers (ints) keyed like the C structs."""
        w = TrainWeights(**{k: (weights.get(k) or None) for k, _ in TrainWeights._fields_})
        o = TrainOutputs(**{k: outputs.get(k) for k, _ in TrainOutputs._fields_})
        check(load().farnn_decomp_ifst_train_step(self._raw, C.byref(w), x_ptr, len_ptr, labels_ptr, int(B), int(L),
                                                  int(valid_tokens), C.byref(o), stream),
              'farnn_decomp_ifst_train_step')

    def set_profiling(self, enable):
        check(load().farnn_train_set_profiling(self._raw, int(enable)), 'farnn_train_set_profiling')

    def time(self):
        ms, n = C.c_double(0), C.c_int64(0)
        check(load().farnn_train_time(self._raw, C.byref(ms), C.byref(n)), 'farnn_train_time')
        return ms.value, n.value

"""ctypes binding of libfarnn_rccl.so (include/farnn_rccl.h): the tag gather over RCCL without torch.distributed.

Optional: the package's default multi-GPU path is `dist.gather_tags_balanced` (torch.distributed, backend "nccl" = RCCL); this
module is the same collective for hosts that own their process group themselves.  Load it after `import torch` (like
libfarnn_hip.so: one HIP runtime per process)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FARNN_RCCL_LIB', os.path.join(HERE, 'csrc', 'libfarnn_rccl.so'))
ID_BYTES = 128
SIGNATURES = {
    'farnn_rccl_unique_id': (C.c_int, [C.c_void_p]),
    'farnn_rccl_comm_create': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'farnn_rccl_gather_tags': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    'farnn_rccl_comm_count': (C.c_int, [C.c_void_p]),
    'farnn_rccl_comm_destroy': (C.c_int, [C.c_void_p]),
    'farnn_rccl_version': (C.c_int, []),
    'farnn_rccl_last_error': (C.c_char_p, []),
}
_lib = None


class RcclError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RcclError('libfarnn_rccl.so is not built: run __graft_entry__.build()')
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _check(rc):
    if rc != 0:
        raise RcclError(lib().farnn_rccl_last_error().decode() or 'error %d' % rc)


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    _check(lib().farnn_rccl_unique_id(buf))
    return buf.raw


class Communicator:
    """One rank's end of an RCCL communicator (collective constructor: returns when all `nranks` ranks have joined)."""

    def __init__(self, uid, nranks, rank, device):
        assert len(uid) == ID_BYTES
        h = C.c_void_p()
        _check(lib().farnn_rccl_comm_create(C.create_string_buffer(uid, ID_BYTES), nranks, rank, device, C.byref(h)))
        self._h, self.nranks, self.rank = h, nranks, rank

    def gather_tags(self, local, out=None, stream=None):
        """local: CUDA int32 [rows, L] (the same `rows` on every rank); returns [nranks * rows, L], rank-major."""
        import torch
        assert local.is_cuda and local.dtype == torch.int32 and local.is_contiguous()
        rows, L = local.shape
        if out is None:
            out = torch.empty((self.nranks * rows, L), dtype=torch.int32, device=local.device)
        s = torch.cuda.current_stream(local.device).cuda_stream if stream is None else stream
        _check(lib().farnn_rccl_gather_tags(self._h, C.c_void_p(local.data_ptr()), rows, L, C.c_void_p(out.data_ptr()), C.c_void_p(s)))
        return out

    def count(self):
        """the number of ranks RCCL reports for this communicator (ncclCommCount)"""
        n = lib().farnn_rccl_comm_count(self._h)
        if n < 0:
            _check(n)
        return n

    def close(self):
        if self._h:
            _check(lib().farnn_rccl_comm_destroy(self._h))
            self._h = None

    def __del__(self):                       # (a communicator that was never closed: destroyed with its owner, errors swallowed)
        try:
            self.close()
        except Exception:
            pass

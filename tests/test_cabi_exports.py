"""The C-ABI library loads on a CPU-only machine and exports every symbol include/farnn.h
declares; no compute entry point is called here (there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, 'include', 'farnn.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(farnn_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    from re2nn_seq_amd import _lib
    assert _declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from re2nn_seq_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.farnn_abi_version.restype = ctypes.c_int
    assert lib.farnn_abi_version() == 2


def test_no_torch_types_in_the_abi():
    with open(os.path.join(ROOT, 'include', 'farnn.h')) as f:
        text = f.read()
    assert '#include <torch' not in text and 'at::' not in text and 'std::' not in text


def test_product_path_fails_loudly_without_a_gpu():
    """No silent CPU fallback: without a device, creating a model raises."""
    import numpy as np
    import torch
    from re2nn_seq_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    T = np.zeros((3, 2, 2), np.float32)
    with pytest.raises(_lib.FarnnError):
        _lib.create_onehot_ifst(T, np.zeros((2, 2)), np.zeros((2, 2)), np.zeros(2), np.zeros(2))
    from re2nn_seq_amd.farnn.model_onehot import FARNN_S_O_I_S
    from util import ns
    m = FARNN_S_O_I_S(T, np.zeros((2, 2)), np.zeros((2, 2)), np.zeros(2), np.zeros(2), np.zeros(2), None, ns())
    x = torch.zeros((1, 2), dtype=torch.int64)
    with pytest.raises(_lib.FarnnError):
        m.forward_local(x, x, torch.ones(1, dtype=torch.int64), train=False)
    with pytest.raises(_lib.FarnnError):
        m.cpu()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under re2nn-seq_amd/ may reference it."""
    pkg = os.path.join(ROOT, 're2nn-seq_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                with open(os.path.join(dirpath, fn)) as f:
                    text = f.read()
                assert 'import oracle' not in text and 'from oracle' not in text, fn
                assert 'farnn_oracle' not in text, fn


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct the binding mirrors, as gcc sees include/farnn.h, against ctypes."""
    import subprocess
    from re2nn_seq_amd import _lib
    pairs = {
        'farnn_onehot_ifst_desc': _lib.OnehotIfstDesc, 'farnn_onehot_fst4_desc': _lib.OnehotFst4Desc,
        'farnn_onehot_ind1_desc': _lib.OnehotInd1Desc, 'farnn_decomp_ifst_desc': _lib.DecompIfstDesc,
        'farnn_decomp_ind1_desc': _lib.DecompInd1Desc, 'farnn_decomp_fst_desc': _lib.DecompFstDesc,
        'farnn_edge_list': _lib.EdgeList, 'farnn_train_dims': _lib.TrainDims,
        'farnn_train_weights': _lib.TrainWeights, 'farnn_train_outputs': _lib.TrainOutputs,
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "farnn.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s sizeof %zu\\n", "{0}", sizeof({0}));'.format(cname))
        for fname, _ in cls._fields_:              # a trailing underscore in the binding avoids a Python keyword
            lines.append('  printf("%s.%s %zu\\n", "{0}", "{1}", offsetof({0}, {2}));'.format(cname, fname, fname.rstrip('_')))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.rsplit(' ', 1) for line in out.strip().splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname + ' sizeof']) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got['{}.{}'.format(cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_no_kernel_of_the_library_uses_scratch_memory():
    """Round 4 retired the scratch: every kernel's registers fit (the code objects' metadata: private_segment_fixed_size = 0)."""
    import subprocess
    import sys
    from re2nn_seq_amd import _lib
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-readelf'):
        pytest.skip('no llvm-readelf')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'kernel_metadata.py'), _lib.LIB_PATH],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith('#') and last.endswith('with scratch: 0'), last


def test_rccl_gather_library_exports_its_header():
    """include/farnn_rccl.h (the tag gather over RCCL without torch.distributed): every declared symbol is exported by
    libfarnn_rccl.so and bound by re2nn_seq_amd._rccl; nothing is called that needs a GPU."""
    import torch  # noqa: F401  (one HIP runtime per process: the libraries are loaded after torch)
    from re2nn_seq_amd import _rccl
    with open(os.path.join(ROOT, 'include', 'farnn_rccl.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    declared = sorted(set(re.findall(r'\b(farnn_rccl_[a-z0-9_]+)\s*\(', text)))
    assert declared == sorted(_rccl.SIGNATURES)
    if not os.path.exists(_rccl.LIB_PATH):
        pytest.skip('libfarnn_rccl.so not built (optional)')
    try:
        lib = ctypes.CDLL(_rccl.LIB_PATH)
    except OSError as e:
        pytest.skip('librccl.so does not load here: %s' % e)
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize('name,why', [('FARNN_CV_ONE', 'A/B build only'), ('FARNN_CV_STASH', 'A/B build only'), ('FARNN_NODEST', 'A/B build only'),
                                      ('FARNN_CV_WIDE', 'removed'), ('FARNN_DECOMP_OLD', 'removed')])
def test_create_refuses_ab_only_and_removed_switches(name, why, monkeypatch):
    """include/farnn.h: switches are resolved when a handle is created.  The production library refuses the forms that live in the
    A/B build only, and both refuse the switches earlier rounds removed -- at create, with a message that names the switch, before
    any device is touched (so this runs without a GPU); nothing is silently ignored and no farnn_tag call fails later."""
    import numpy as np
    from re2nn_seq_amd import _lib
    if _lib.ab_build() and why != 'removed':
        pytest.skip('the A/B build carries this form')
    monkeypatch.setenv(name, '1')
    T = np.zeros((3, 2, 2), np.float32)
    with pytest.raises(_lib.FarnnError) as e:
        _lib.create_onehot_ifst(T, np.zeros((2, 2)), np.zeros((2, 2)), np.zeros(2), np.zeros(2))
    assert name in str(e.value) and why in str(e.value), str(e.value)

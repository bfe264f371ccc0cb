"""Build libfarnn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python re2nn-seq_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'farnn_hip.hip')
OUT = os.path.join(HERE, 'libfarnn_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC']


def _sources():
    root = os.path.dirname(os.path.dirname(HERE))
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(('.hip', '.h'))]
    srcs.append(os.path.join(root, 'include', 'farnn.h'))
    return srcs


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(s) <= t for s in _sources())


def build_hip(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [HIPCC] + FLAGS + os.environ.get('FARNN_EXTRA_FLAGS', '').split() + ['-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return OUT


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)

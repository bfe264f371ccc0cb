"""bench.py's N > 1 launch contract on a one-GPU box: two ranks started with torch.distributed.run, both on cuda:0
over gloo (FARNN_BENCH_ONE_DEVICE=1; with one rank per device the same code runs over RCCL).  Checks the single JSON
line rank 0 prints."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('extra', [[], ['--workload', 'train']])
def test_bench_two_ranks_on_one_device(extra):
    env = dict(os.environ, FARNN_BENCH_ONE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5',
           '--warmup', '2', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['warmup'] == 2 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['config']['valid_tokens_per_step'] > 8000          # both ranks' shards are counted
    assert 'roofline' in d and d['roofline']['frac'] > 0

"""The onehot i-FST's dispatch rule (csrc/farnn_hip.hip, launch_chain: ONE launch -- scores + decode beside the recurrence -- while
2 B <= compute units, else the recurrence kernel followed by the label-map score launch) was calibrated at B = 64 / 256 / 1 024 and
L = 64 only.  A same-process A/B at the batch sizes in between and at the reference's default `--seq_max_len 30`: whatever form the
library picks by itself must not be slower than the other one by more than 5 % (both forms are held to the same tags)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _time(h, xd, ld, B, L, tags, n=300):
    from re2nn_seq_amd import _lib
    for _ in range(30):
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags.data_ptr(), None, None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


@pytest.mark.parametrize('B,L', [(64, 30), (128, 30), (129, 30), (192, 30), (255, 30), (129, 64), (192, 64), (255, 64), (200, 30)])
def test_the_chosen_form_is_not_the_slower_one(B, L, monkeypatch):
    from re2nn_seq_amd import _lib, synth
    V, S, C = 950, 71, 128
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, np.random.RandomState(1234))
    x, lengths = synth.random_batch(V, B, L, np.random.RandomState(99))
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda()
    handles = {}
    for name, env in (('default', {}), ('one_launch', {'FARNN_FUSE': '1'}), ('two_launches', {'FARNN_NOFUSE': '1'})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        handles[name] = _lib.create_onehot_ifst(T, W, O, h0, hT)      # (switches are read when the handle is created)
        for k in env:
            monkeypatch.delenv(k)
    tags = {k: torch.empty((B, L), dtype=torch.int32, device='cuda') for k in handles}

    def measure(rounds):
        us = {k: [] for k in handles}
        for _ in range(rounds):                          # interleaved rounds: clock drift hits all three alike
            for k, h in handles.items():
                us[k].append(_time(h, xd, ld, B, L, tags[k]))
        return {k: min(v) for k, v in us.items()}
    best = measure(3)
    for k in ('one_launch', 'two_launches'):
        assert torch.equal(tags[k], tags['default']), k
    other = min(best['one_launch'], best['two_launches'])
    if best['default'] > 1.05 * other:                   # (a noisy box: one more, longer look before the verdict)
        again = measure(6)
        best = {k: min(best[k], again[k]) for k in best}
        other = min(best['one_launch'], best['two_launches'])
    for h in handles.values():
        h.close()
    print('B = {} L = {}: default {:.1f} us, one launch {:.1f}, two launches {:.1f}'.format(B, L, best['default'], best['one_launch'], best['two_launches']))
    assert best['default'] <= 1.05 * other, best

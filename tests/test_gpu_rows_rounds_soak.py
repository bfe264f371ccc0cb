"""A short run of tests/soak_rows_rounds.py in the suite: random gated decomposed models on the rows kernel's register forms, batches
around and beyond the slot count, empty sequences and ties -- rounds (default) against FARNN_ROWS_NOROUNDS=1 bit for bit, every tenth
draw against the oracle.  (Its first long run found a wild table row read through a token word that an EMPTY sequence never wrote:
stale LDS of whatever kernel ran before, a memory fault once in a few draws.)"""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_rounds_agree_with_one_workgroup_per_chain_over_random_shapes():
    import soak_rows_rounds
    n = int(os.environ.get('FARNN_SHAPE_SOAK', '0')) or 60
    assert soak_rows_rounds.run(n, seed=int(os.environ.get('FARNN_SHAPE_SEED', '0'))) == 0

#!/usr/bin/env python3
"""Golden fixture for a rule word that is NOT in the vocabulary: the reference's loaders
(src_seq/wfa/fsa_to_tensor.py:398-615) write the edge's LABEL before they look the word up, so output_mat
(independent=2) and output_tensor (independent=1) keep the label while the language tensor skips the edge;
the 4-D layout drops it.  Runs only in the build container (imports the reference loader from /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_oov.py

Input: the automaton of loader_small.json plus the edges listed in OOV_EDGES (also applied by the test).
Output: tests/golden/loader_oov.npz (data only)."""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
from src_seq.wfa import fsa_to_tensor as ref_f2t  # noqa: E402

# (from, to, 'word<:>slot'): a literal OOV word with a slot label no other edge into that state carries
OOV_EDGES = [(0, 1, 'notaword<:>b-e1'), (2, 3, 'alsomissing<:>i-e2'), (1, 1, 'nope<:>oo')]


def automaton_with_oov():
    with open(os.path.join(HERE, 'loader_small.json')) as f:
        meta = json.load(f)
    a = meta['automaton']
    automaton = {'states': set(a['states']), 'startstate': a['startstate'], 'finalstates': a['finalstates'],
                 'transitions': {int(fr): {int(to): set(e) for to, e in d.items()}
                                 for fr, d in a['transitions'].items()}}
    for fr, to, lab in OOV_EDGES:
        automaton['transitions'].setdefault(fr, {}).setdefault(to, set()).add(lab)
    return automaton, meta['t2i'], meta['s2i']


if __name__ == '__main__':
    automaton, t2i, s2i = automaton_with_oov()
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        r = ref_f2t.dfa_to_tensor_slot_independent_wildcard(automaton, t2i, s2i)
        out.update({'ind.T': r[0], 'ind.W': r[2], 'ind.Oten': r[3]})
        r = ref_f2t.dfa_to_tensor_slot_single_wildcard(automaton, t2i, s2i)
        out.update({'single.T': r[0], 'single.W': r[2], 'single.O': r[3]})
        # the 4-D loader asserts word == '$' on 'oo' edges, so it gets the automaton without the 'oo' OOV edge
        a4 = {**automaton, 'transitions': {f: {t: {e for e in es if e != 'nope<:>oo'} for t, es in d.items()}
                                           for f, d in automaton['transitions'].items()}}
        r = ref_f2t.dfa_to_tensor_slot_new_wildcard(a4, t2i, s2i)
        out.update({'new.T4': r[0], 'new.W4': r[2]})
    np.savez_compressed(os.path.join(HERE, 'loader_oov.npz'), **out)
    print('wrote loader_oov.npz', {k: v.shape for k, v in out.items()})

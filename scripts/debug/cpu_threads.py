"""The C port's throughput form by thread count, with and without OpenMP thread binding (host-only; what bench.py's cpu_baseline sweeps)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_port
from re2nn_seq_amd import synth
c_port.load(native=True)
rng = np.random.RandomState(1234)
T, W, O, h0, hT = synth.random_ifst_tensors(950, 71, 128, rng)
x, l = synth.random_batch(950, 256, 64, np.random.RandomState(4321))
Tf = T + W
tok = int(l.sum())
for nt in [int(v) for v in sys.argv[1:]] or [16, 32, 64, 128, 256]:
    c_port.onehot_ifst_tag(Tf, O, h0, hT, x, l, nthreads=nt, reps=4, stream=True)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        c_port.onehot_ifst_tag(Tf, O, h0, hT, x, l, nthreads=nt, reps=200, stream=True); n += 200
    el = time.perf_counter() - t0
    print('threads %3d: %.3e tok/s (%s)' % (nt, tok * n / el, os.environ.get('OMP_PROC_BIND', 'unbound')), flush=True)

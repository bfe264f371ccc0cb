"""re2nn-seq_amd: MI355X-native forward tagging path of the FA-RNN slot tagger.

Only what the hot path needs lives here:
  csrc/     hand-written HIP kernels for gfx950 + the C-ABI (include/farnn.h)
  _lib.py   ctypes binding of that C-ABI (fails loudly when the library is missing)
  farnn/    host-side mirror of the reference's model classes (same names/methods)
  wfa/      automaton dict -> dense tensors (the ".pkl loader" boundary)
  ...       init_params / data / val / RE / main: the callers either side of the path
"""
__version__ = "0.1.0"

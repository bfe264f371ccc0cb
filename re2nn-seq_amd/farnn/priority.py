"""Priority matrix of the label scores (reference src_seq/farnn/priority.py).

The reference applies ``scores @ priority_mat + 0`` as an nn.Module; here the expanded matrix is
handed to the HIP library, which applies it in the scoring epilogue (K7 in SURVEY.md 2b).
"""
import numpy as np


def expand_priority(C, priority_mat=None):
    """identity[C,C] with `priority_mat` copied into the top-left corner (ref :6-18)."""
    full = np.eye(C, dtype=np.float32)
    if priority_mat is not None:
        pm = np.asarray(priority_mat, dtype=np.float32)
        n = pm.shape[0]
        full[:n, :n] = pm
    return full


class PriorityLayer:
    """Name-compatible holder of the expanded matrix (ref :5-30)."""

    def __init__(self, C, priority_mat=None, priority_bias=None):
        self.priority_mat = expand_priority(C, priority_mat)
        self.priority_bias = np.zeros(C, dtype=np.float32)

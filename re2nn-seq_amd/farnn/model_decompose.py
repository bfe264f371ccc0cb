"""Host mirror of the reference's 4th-order decomposed tagger ``FARNN_S_D_W``
(src_seq/farnn/model_decompose.py:10-459; ``--independent 0``): language tensor
T4[v,c,s,j] ~ sum_r V C S1 S2, wildcard tensor W4[c,s,j] ~ sum_q C_wildcard S1_wildcard S2_wildcard,
plus the dense wildcard_wildcard matrix.  Same constructor arguments and state-dict keys as the
reference; inference runs in the HIP library (farnn_decomp_fst_create).

The reference's own CLI cannot reach this class (main.py:127 forces local_loss_func='CE1' and
train_decompose.py:69-70 asserts independent != 0 under CE1); it is reachable as a Python class.
"""
import numpy as np
import torch

from .. import _lib
from .model_decompose_single import FARNN_S_D_W_I_S, crf_default_transitions, _GATE_KEYS
from ._native import NativeTagger
from .priority import expand_priority

_PARAM_KEYS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_embed', 'C_wildcard', 'S1_wildcard',
               'S2_wildcard', 'wildcard_wildcard', 'h0', 'hT', 'beta_vec') + _GATE_KEYS


class FARNN_S_D_W(FARNN_S_D_W_I_S):
    _param_keys = _PARAM_KEYS

    def __init__(self, V=None, C=None, S1=None, S2=None, C_wildcard=None, S1_wildcard=None,
                 S2_wildcard=None, wildcard_wildcard=None, final_vector=None, start_vector=None,
                 pretrained_word_embed=None, priority_mat=None, args=None, o_idx=0, is_cuda=True):
        NativeTagger.__init__(self, args, o_idx)
        self.additional_states = int(args.additional_states)
        self.embedding = torch.from_numpy(np.asarray(pretrained_word_embed)).float()       # V x D
        self.C, self.R_W = C_wildcard.shape                                                # (:38-40)
        self.S, _ = S1_wildcard.shape
        _, self.R = C.shape
        self.use_crf = bool(args.use_crf)
        self.crf_transitions = None
        if self.use_crf:
            self.crf_transitions = crf_default_transitions(self.C)
            self.C += 2
        self.priority_full = expand_priority(self.C, priority_mat)
        self.random = bool(args.random)
        self.h0 = self.pad_additional_states(torch.from_numpy(np.asarray(start_vector)).float())
        self.hT = self.pad_additional_states(torch.from_numpy(np.asarray(final_vector)).float())
        self._init_forward_parameters_fst(S1, S2, C, V, S1_wildcard, S2_wildcard, C_wildcard,
                                          wildcard_wildcard)
        self.beta = args.beta
        self.beta_vec = torch.tensor([self.beta] * self.R).float()

    # ---- parameter construction (ref model_decompose.py:104-181) -------------------------------
    def _init_forward_parameters_fst(self, S1, S2, C, V, S1_w, S2_w, C_w, W):
        a = self.args
        t = lambda x: torch.from_numpy(np.asarray(x)).float()      # noqa: E731
        self.S1 = self.pad_additional_states(t(S1))
        self.S2 = self.pad_additional_states(t(S2))
        self.V_embed = t(V)
        self.embed_r_generalized = torch.matmul(self.embedding.pinverse(), self.V_embed)   # D x R (:108-111)
        C, C_w = np.asarray(C), np.asarray(C_w)
        if a.use_crf == 1:       # two extra rows for START/STOP on both label factors (:114-116)
            C = np.concatenate((C, self.get_random((2, self.R)).numpy() * a.rand_constant), axis=0)
            C_w = np.concatenate((C_w, self.get_random((2, self.R_W)).numpy() * a.rand_constant), axis=0)
        self.C_wildcard = self.pad_additional_states(t(C_w))
        self.C_embed = t(C)                                        # not state-padded (:122)
        self.S1_wildcard = self.pad_additional_states(t(S1_w))
        self.S2_wildcard = self.pad_additional_states(t(S2_w))
        self.wildcard_wildcard = self.pad_additional_states(t(W))
        Sp = self.S + self.additional_states
        if a.farnn in (1, 2):    # gate parameters (:138-165)
            self.Wss1 = torch.randn((Sp, Sp)).float()
            self.Wrs1 = torch.randn((self.R, Sp)).float()
            self.bs1 = torch.ones((1, Sp)).float() * a.bias_init
            if a.farnn == 2:
                self.Wss2 = torch.randn((Sp, Sp)).float()
                self.Wrs2 = torch.randn((self.R, Sp)).float()
                self.bs2 = torch.ones((1, Sp)).float() * a.bias_init
            if a.xavier:
                for name in _GATE_KEYS:
                    if hasattr(self, name) and not name.startswith('bs'):
                        torch.nn.init.xavier_normal_(getattr(self, name))
                if a.farnn == 1:
                    torch.nn.init.xavier_normal_(self.bs1)
        if self.random:          # (:167-181)
            for name in ('S1', 'S2', 'C_embed', 'V_embed', 'S1_wildcard', 'S2_wildcard', 'C_wildcard',
                         'embed_r_generalized', 'wildcard_wildcard'):
                torch.nn.init.xavier_normal_(getattr(self, name))
            torch.nn.init.normal_(self.h0)
            torch.nn.init.normal_(self.hT)

    # ---- device handle ------------------------------------------------------------------------
    def _build_handle(self):
        a = self.args
        gates = {k: getattr(self, k).reshape(-1) if k.startswith('bs') else getattr(self, k)
                 for k in _GATE_KEYS if hasattr(self, k)}
        gates = {k: v.numpy() for k, v in gates.items()}
        return _lib.create_decomp_fst(
            self.generalized_vocab_table().numpy(), self.C_embed.numpy(), self.S1.numpy(), self.S2.numpy(),
            self.C_wildcard.numpy(), self.S1_wildcard.numpy(), self.S2_wildcard.numpy(),
            self.wildcard_wildcard.numpy(), self.h0.numpy(), self.hT.numpy(),
            P=self.priority_full if a.use_priority else None, farnn=a.farnn, gates=gates,
            sigmoid_exponent=a.sigmoid_exponent, nl=a.update_nonlinear,
            semiring='max' if a.train_mode == 'max' else 'sum', threshold=a.threshold,
            o_idx=self.o_idx, use_crf=self.use_crf,
            crf_trans=None if self.crf_transitions is None else self.crf_transitions.numpy(),
            device=self.device_index)

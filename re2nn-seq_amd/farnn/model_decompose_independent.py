"""Host mirror of the reference's decomposed independent=1 tagger ``FARNN_S_D_W_I``
(src_seq/farnn/model_decompose_independent.py:11-300): a rank-R language tensor plus a separate
rank-R_O output tensor (C_output, S1_output, S2_output).  Same constructor arguments and state-dict
keys as the reference; inference runs in the HIP library (farnn_decomp_ind1_create).
"""
import numpy as np
import torch

from .. import _lib
from .model_decompose_single import FARNN_S_D_W_I_S, crf_default_transitions, _GATE_KEYS
from ._native import NativeTagger
from .priority import expand_priority

_PARAM_KEYS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_output', 'S1_output', 'S2_output',
               'wildcard_mat', 'wildcard_output', 'h0', 'hT', 'beta_vec') + _GATE_KEYS


class FARNN_S_D_W_I(FARNN_S_D_W_I_S):
    _param_keys = _PARAM_KEYS

    def __init__(self, V=None, S1=None, S2=None, C_output=None, S1_output=None, S2_output=None,
                 wildcard_mat=None, wildcard_output=None, final_vector=None, start_vector=None,
                 pretrained_word_embed=None, priority_mat=None, args=None, o_idx=0, is_cuda=True):
        NativeTagger.__init__(self, args, o_idx)
        self.additional_states = int(args.additional_states)
        self.embedding = torch.from_numpy(np.asarray(pretrained_word_embed)).float()       # V x D
        self.C, self.R_O = C_output.shape                                                  # (:39)
        self.S, self.R = S1.shape
        self.use_crf = bool(args.use_crf)
        self.crf_transitions = None
        if self.use_crf:
            self.crf_transitions = crf_default_transitions(self.C)
            self.C += 2
        self.priority_full = expand_priority(self.C, priority_mat)
        self.random = bool(args.random)
        self.h0 = self.pad_additional_states(torch.from_numpy(np.asarray(start_vector)).float())
        self.hT = self.pad_additional_states(torch.from_numpy(np.asarray(final_vector)).float())
        self._init_forward_parameters_ind1(S1, S2, V, S1_output, S2_output, C_output, wildcard_mat,
                                           wildcard_output)
        self.beta = args.beta
        self.beta_vec = torch.tensor([self.beta] * self.R).float()

    # ---- parameter construction (ref model_decompose_independent.py:68-146) ---------------------
    def _init_forward_parameters_ind1(self, S1, S2, V, S1_o, S2_o, C_o, W, W_o):
        a = self.args
        t = lambda x: torch.from_numpy(np.asarray(x)).float()      # noqa: E731
        self.S1 = self.pad_additional_states(t(S1))
        self.S2 = self.pad_additional_states(t(S2))
        self.V_embed = t(V)
        self.embed_r_generalized = torch.matmul(self.embedding.pinverse(), self.V_embed)   # D x R (:72-75)
        C_o = np.asarray(C_o)
        if a.use_crf:            # two extra rows for START/STOP (:77-79)
            C_o = np.concatenate((C_o, self.get_random((2, self.R_O)).numpy() * a.rand_constant), axis=0)
        self.C_output = self.pad_additional_states(t(C_o))
        self.S1_output = self.pad_additional_states(t(S1_o))
        self.S2_output = self.pad_additional_states(t(S2_o))
        self.wildcard_mat = self.pad_additional_states(t(W))
        if W_o is not None:
            self.wildcard_output = self.pad_additional_states(t(W_o))
        Sp = self.S + self.additional_states
        if a.farnn in (1, 2):    # gate parameters (:99-130)
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
        if self.random:          # (:132-146)
            for name in ('S1', 'S2', 'V_embed', 'S1_output', 'S2_output', 'C_output',
                         'embed_r_generalized', 'wildcard_mat'):
                torch.nn.init.xavier_normal_(getattr(self, name))
            torch.nn.init.normal_(self.h0)
            torch.nn.init.normal_(self.hT)

    # ---- device handle ------------------------------------------------------------------------
    def _build_handle(self):
        a = self.args
        gates = {k: getattr(self, k).reshape(-1) if k.startswith('bs') else getattr(self, k)
                 for k in _GATE_KEYS if hasattr(self, k)}
        gates = {k: v.numpy() for k, v in gates.items()}
        Wo = None                      # CE1 drops wildcard_output from the output sum (:213-216)
        if a.local_loss_func != 'CE1' and getattr(self, 'wildcard_output', None) is not None:
            Wo = self.wildcard_output.numpy()
        return _lib.create_decomp_ind1(
            self.generalized_vocab_table().numpy(), self.S1.numpy(), self.S2.numpy(),
            self.wildcard_mat.numpy(), self.C_output.numpy(), self.S1_output.numpy(),
            self.S2_output.numpy(), self.h0.numpy(), self.hT.numpy(), Wo=Wo,
            P=self.priority_full if a.use_priority else None, farnn=a.farnn, gates=gates,
            sigmoid_exponent=a.sigmoid_exponent, nl=a.update_nonlinear,
            semiring='max' if a.train_mode == 'max' else 'sum', threshold=a.threshold,
            o_idx=self.o_idx, use_crf=self.use_crf,
            crf_trans=None if self.crf_transitions is None else self.crf_transitions.numpy(),
            device=self.device_index)

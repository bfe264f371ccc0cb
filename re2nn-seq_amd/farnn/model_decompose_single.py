"""Host mirror of the reference's decomposed i-FST tagger ``FARNN_S_D_W_I_S``
(src_seq/farnn/model_decompose_single.py:12-304; shared pieces from model_decompose.py).

The constructor keeps the reference's argument list and builds the same parameter set
(state padding with ``additional_states``, the pseudo-inverse word-embedding bridge, optional
GRU-style gates, optional CRF) with torch CPU ops; inference runs in the HIP library.  Because the
weights are frozen on the tagging path, the per-token ``get_generalized_v_embed_vec``
(model_decompose.py:222-241) is folded once into a table ``Vgen[V,R]`` when the device handle is
built -- the reference recomputes it twice per time step.
"""
import numpy as np
import torch

from .. import _lib
from ._native import NativeTagger
from .priority import expand_priority

_GATE_KEYS = ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')
_PARAM_KEYS = ('S1', 'S2', 'V_embed', 'embed_r_generalized', 'C_output_mat', 'wildcard_mat',
               'wildcard_output_vector', 'h0', 'hT', 'beta_vec') + _GATE_KEYS


def crf_default_transitions(tagset_size):
    """CRF.__init__ of the reference (baselines/crf.py:31-46)."""
    K = tagset_size + 2
    tr = torch.zeros(K, K)
    tr[:, K - 2] = -10000.0
    tr[K - 1, :] = -10000.0
    return tr


class FARNN_S_D_W_I_S(NativeTagger):
    _param_keys = _PARAM_KEYS
    local_uses_max_len = True        # forward_local iterates lengths.max() positions (ref :221)

    def __init__(self, V=None, S1=None, S2=None, C_output_mat=None, wildcard_mat=None,
                 wildcard_output_vector=None, final_vector=None, start_vector=None,
                 pretrained_word_embed=None, priority_mat=None, args=None, o_idx=0, is_cuda=True):
        super().__init__(args, o_idx)
        self.additional_states = int(args.additional_states)
        self.embedding = torch.from_numpy(np.asarray(pretrained_word_embed)).float()       # V x D
        self.C = C_output_mat.shape[0]
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
        self._init_forward_parameters(S1, S2, V, C_output_mat, wildcard_mat, wildcard_output_vector)
        self.beta = args.beta
        self.beta_vec = torch.tensor([self.beta] * self.R).float()

    # ---- parameter construction (ref model_decompose_single.py:68-136) ------------------------
    def get_random(self, sizes):
        """ref model_decompose.py:189-199"""
        f = self.args.random_pad_func
        if f == 'uniform':
            return torch.rand(sizes)
        if f == 'normal':
            return torch.randn(sizes)
        a = torch.randn(sizes)
        torch.nn.init.xavier_normal_(a)
        return a

    def pad_additional_states(self, obj):
        """ref model_decompose.py:201-220: EVERY dimension equal to S grows by additional_states;
        new entries are rand * rand_constant (zeros for vectors)."""
        shape = tuple(obj.shape)
        padded = tuple(d if d != self.S else d + self.additional_states for d in shape)
        if len(shape) == 1:
            out = torch.zeros(padded)
        else:
            out = self.get_random(padded) * self.args.rand_constant
        out[tuple(slice(0, d) for d in shape)] = obj
        return out

    def _init_forward_parameters(self, S1, S2, V, C_o, W, W_o):
        a = self.args
        t = lambda x: torch.from_numpy(np.asarray(x)).float()      # noqa: E731
        self.S1 = self.pad_additional_states(t(S1))
        self.S2 = self.pad_additional_states(t(S2))
        self.V_embed = t(V)
        self.embed_r_generalized = torch.matmul(self.embedding.pinverse(), self.V_embed)   # D x R (:73-76)
        C_o = np.asarray(C_o)
        if a.use_crf == 1:       # two extra rows for START/STOP (:78-79)
            C_o = np.concatenate((C_o, self.get_random((2, self.S)).numpy() * a.rand_constant), axis=0)
        self.C_output_mat = self.pad_additional_states(t(C_o))
        self.wildcard_mat = self.pad_additional_states(t(W))
        self.wildcard_output_vector = self.pad_additional_states(t(W_o))
        Sp = self.S + self.additional_states
        if a.farnn in (1, 2):    # gate parameters (:93-123)
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
        if self.random:          # (:125-136)
            for name in ('S1', 'S2', 'V_embed', 'C_output_mat', 'embed_r_generalized', 'wildcard_mat'):
                torch.nn.init.xavier_normal_(getattr(self, name))
            torch.nn.init.normal_(self.h0)
            torch.nn.init.normal_(self.hT)

    # ---- state dict compatible with the reference's key names ---------------------------------
    def state_dict(self):
        sd = {k: getattr(self, k) for k in self._param_keys if hasattr(self, k)}
        sd['embedding.weight'] = self.embedding
        sd['priority_layer.priority_mat'] = torch.from_numpy(self.priority_full)
        if self.use_crf:
            sd['crf.transitions'] = self.crf_transitions
        return sd

    def load_state_dict(self, sd, strict=False):
        for k, v in sd.items():
            v = torch.as_tensor(np.asarray(v)).float() if not torch.is_tensor(v) else v.detach().float().cpu()
            if k in self._param_keys:
                setattr(self, k, v)
            elif k == 'embedding.weight':
                self.embedding = v
            elif k == 'priority_layer.priority_mat':
                self.priority_full = v.numpy()
            elif k == 'crf.transitions':
                self.crf_transitions = v
        self.invalidate()
        return self

    # ---- device handle ------------------------------------------------------------------------
    def generalized_vocab_table(self):
        """get_generalized_v_embed_vec for every word id at once (ref model_decompose.py:222-241)."""
        gen = torch.matmul(self.embedding, self.embed_r_generalized)
        nl = self.args.additional_nonlinear
        if nl == 'relu':
            gen = torch.relu(gen)
        elif nl == 'tanh':
            gen = torch.tanh(gen)
        elif nl == 'sigmoid':
            gen = torch.sigmoid(gen)
        elif nl == 'relutanh':
            gen = torch.tanh(torch.relu(gen))
        return self.V_embed * self.beta_vec + gen * (1 - self.beta_vec)

    def _build_handle(self):
        a = self.args
        if a.local_loss_func != 'CE1':
            raise NotImplementedError('only CE1 is reachable from main.py (:127)')
        gates = {k: getattr(self, k).reshape(-1) if k.startswith('bs') else getattr(self, k)
                 for k in _GATE_KEYS if hasattr(self, k)}
        gates = {k: v.numpy() for k, v in gates.items()}
        # the word table Vgen = V_embed * beta + nl_add(E @ G) * (1 - beta) is folded by the library on the device
        # (farnn_decomp_ifst_create_folded); generalized_vocab_table() remains as the host statement of it
        return _lib.create_decomp_ifst_folded(
            self.V_embed.numpy(), self.embedding.numpy(), self.embed_r_generalized.numpy(), self.beta_vec.numpy(),
            self.S1.numpy(), self.S2.numpy(),
            self.wildcard_mat.numpy(), self.C_output_mat.numpy(), self.h0.numpy(), self.hT.numpy(),
            add_nl=a.additional_nonlinear,
            P=self.priority_full if a.use_priority else None, farnn=a.farnn, gates=gates,
            sigmoid_exponent=a.sigmoid_exponent, nl=a.update_nonlinear,
            semiring='max' if a.train_mode == 'max' else 'sum', threshold=a.threshold,
            o_idx=self.o_idx, use_crf=self.use_crf,
            crf_trans=None if self.crf_transitions is None else self.crf_transitions.numpy(),
            device=self.device_index)

    def forward_RE(self, input, label, lengths, train=False):
        raise NotImplementedError('forward_RE exists only on the onehot models (ref model_onehot.py:148)')

    # ---- training step (reference :207-304 with train=True + train_decompose.py:186-190) ----
    # which tensors get a gradient: the reference's requires_grad flags (model_decompose.py:53-63,105-132)
    _TRAIN_FLAGS = {'S1': None, 'S2': None, 'embed_r_generalized': None, 'V_embed': 'train_V_embed',
                    'C_output_mat': 'train_c_output', 'wildcard_mat': 'train_wildcard', 'h0': 'train_h0',
                    'hT': 'train_hT', 'beta_vec': 'train_beta', 'embedding.weight': 'train_word_embed'}

    def _check_trainable(self, re_tags):
        a = self.args
        if a.farnn not in (0, 1, 2) or a.train_mode != 'sum' or a.local_loss_func != 'CE1' or re_tags is not None \
                or getattr(a, 'marryup_type', 'none') not in ('none', None):
            raise NotImplementedError('the HIP training step covers the sum semiring and the CE1 loss (with or without '
                                      'the CRF), no KD/PR teachers (DESIGN.md, row f3)')

    def enable_training(self):
        """Device-resident leaf tensors for the optimizer (returned by parameters()) and the library context."""
        if getattr(self, '_tp', None) is not None:
            return self
        if not torch.cuda.is_available():
            raise _lib.FarnnError('no MI355X visible; the training step has no CPU fallback')
        dev = self._dev()
        src = dict(self.state_dict())
        self._tp = {}
        for k, flag in self._TRAIN_FLAGS.items():
            t = src[k].detach().to(dev).float().clone()
            t.requires_grad_(True if flag is None else bool(getattr(self.args, flag, 0)))
            self._tp[k] = t
        for k in _GATE_KEYS[:3 * int(self.args.farnn)]:
            self._tp[k] = src[k].detach().to(dev).float().clone().requires_grad_(True)
        if self.use_crf:
            self._tp['crf.transitions'] = self.crf_transitions.detach().to(dev).float().clone().requires_grad_(True)
        self._tpP = torch.from_numpy(np.ascontiguousarray(self.priority_full, dtype=np.float32)).to(dev) \
            if self.args.use_priority else None
        S, R = self._tp['S1'].shape
        self._tc = _lib.TrainContext(self._tp['V_embed'].shape[0], S, R, self._tp['C_output_mat'].shape[0],
                                     nl=self.args.update_nonlinear, threshold=self.args.threshold, o_idx=self.o_idx,
                                     device=self.device_index, use_crf=self.use_crf, farnn=int(self.args.farnn),
                                     sigmoid_exponent=float(self.args.sigmoid_exponent))
        self._dirty = False
        return self

    def parameters(self):
        tp = getattr(self, '_tp', None)
        return iter(()) if tp is None else iter([t for t in tp.values() if t.requires_grad])

    def named_parameters(self):
        tp = getattr(self, '_tp', None)
        return iter(()) if tp is None else iter([(k, t) for k, t in tp.items() if t.requires_grad])

    def sync_from_training(self):
        """Copy the trained tensors back into the host attributes the tagging handle is built from."""
        tp = getattr(self, '_tp', None)
        if tp is None or not self._dirty:
            return
        for k, t in tp.items():
            if k == 'embedding.weight':
                self.embedding = t.detach().cpu()
            elif k == 'crf.transitions':
                self.crf_transitions = t.detach().cpu()
            else:
                setattr(self, k, t.detach().cpu())
        self._dirty = False
        self.invalidate()

    def eval(self):
        self.sync_from_training()
        return super().eval()

    def _train_vgen(self):
        tp = self._tp
        gen = torch.matmul(tp['embedding.weight'], tp['embed_r_generalized'])
        nl = self.args.additional_nonlinear
        if nl == 'relu':
            gen = torch.relu(gen)
        elif nl == 'tanh':
            gen = torch.tanh(gen)
        elif nl == 'sigmoid':
            gen = torch.sigmoid(gen)
        elif nl == 'relutanh':
            gen = torch.tanh(torch.relu(gen))
        return tp['V_embed'] * tp['beta_vec'] + gen * (1 - tp['beta_vec'])

    def forward_local(self, input, label, lengths, train=True, re_tags=None):
        if not train:
            self.sync_from_training()
            return super().forward_local(input, label, lengths, train=False, re_tags=re_tags)
        from .train_step import decomp_ifst_train_step
        self._check_trainable(re_tags)
        self.enable_training()
        tp = self._tp
        Lmax = int(lengths.max().item())
        x = input[:, :Lmax]
        lab = label[:, :Lmax]
        loss, tags = decomp_ifst_train_step(self._tc, self._train_vgen(), tp['S1'], tp['S2'], tp['wildcard_mat'],
                                            tp['C_output_mat'], tp['h0'], tp['hT'], self._tpP, x, lengths, lab,
                                            crf_trans=tp.get('crf.transitions'),
                                            gates=tuple(tp[k] for k in _GATE_KEYS[:3 * int(self.args.farnn)]))
        self._dirty = True
        pred = self._flatten(tags, lengths.to(tags.device)).to(torch.int64).to(input.device)
        true = self._flatten(label, lengths).to(input.device)
        return loss, pred, true

"""Host mirrors of the reference's onehot FA-RNN taggers (src_seq/farnn/model_onehot.py).

Same class names, constructor arguments and inference methods as the reference, so
``train_onehot.py``-style drivers construct and call them unchanged; all arithmetic runs in the
HIP library through the C-ABI (include/farnn.h).

  FARNN_S_O      FST, 4-D tensor          (--independent 0)   ref :8-181
  FARNN_S_O_I    two 3-D tensors          (--independent 1)   ref :184-306
  FARNN_S_O_I_S  i-FST, T[V,S,S] + O[C,S] (--independent 2)   ref :310-428
"""
import numpy as np
import torch

from .. import _lib
from ._native import NativeTagger
from .priority import expand_priority


def _noisy(arr, amp):
    """reference utils.add_random_noise (:273-274) applied after the float32 cast."""
    t = torch.from_numpy(np.asarray(arr)).float()
    if amp:
        t = t + torch.rand_like(t) * amp
    return t.numpy()


class _OnehotBase(NativeTagger):
    def __init__(self, args, o_idx, n_labels, priority_mat, is_cuda=False):
        super().__init__(args, o_idx)
        if args.local_loss_func not in ('CE', 'CE1'):
            raise NotImplementedError()            # ref :63-64
        self.C = int(n_labels)
        self.amp = args.rand_constant
        self.priority_full = expand_priority(self.C, priority_mat)

    def _P(self):
        return self.priority_full if self.args.use_priority else None


class FARNN_S_O_I_S(_OnehotBase):
    def __init__(self, language_tensor=None, output_mat=None, wildcard_mat=None,
                 output_wildcard_vector=None, final_vector=None, start_vector=None, priority_mat=None,
                 args=None, o_idx=0, is_cuda=False):
        C, S = output_mat.shape
        super().__init__(args, o_idx, C, priority_mat, is_cuda)
        self.S = S
        # same order of (possibly noisy) casts as the reference constructor (:322-336)
        self.h0 = _noisy(start_vector, self.amp)
        self.hT = _noisy(final_vector, self.amp)
        self.language_tensor = _noisy(language_tensor, self.amp)
        self.wildcard_mat = _noisy(wildcard_mat, self.amp)
        self.output_mat = _lib.f32(output_mat)
        self.output_wildcard_vector = _lib.f32(output_wildcard_vector)
        self.use_crf = False          # the reference's onehot models never read use_crf
        self.crf_transitions = None

    @classmethod
    def from_automaton(cls, automata, word2idx, slot2idx, priority_mat=None, args=None, o_idx=0,
                       dataset='MITR-BIO'):
        """The same tagger built WITHOUT the dense host tensors: the automaton's edges go
        to the device as int32 lists and are scattered in HBM (farnn_onehot_ifst_create_from_edges).
        Only for rand_constant == 0 (the noise of utils.add_random_noise is dense by nature)."""
        from ..wfa.fsa_to_tensor import dfa_to_edges_slot_single_wildcard
        if args.rand_constant:
            raise ValueError('from_automaton needs rand_constant == 0')
        word, frm, to, label, fin, sta, _ = dfa_to_edges_slot_single_wildcard(automata, word2idx, slot2idx, dataset)
        self = cls.__new__(cls)
        C, S = len(slot2idx) + 1, len(automata['states'])
        _OnehotBase.__init__(self, args, o_idx, C, priority_mat, False)
        self.S, self.V = S, len(word2idx)
        self.h0, self.hT = _noisy(sta, 0), _noisy(fin, 0)
        self.edges = (word, frm, to, label)
        self.use_crf = False
        self.crf_transitions = None
        return self

    def _dense(self):
        """language_tensor / wildcard_mat / output_mat of an edge-built model, on demand (state_dict)."""
        word, frm, to, label = self.edges
        C = self.C
        T = np.zeros((self.V, self.S, self.S), np.float32)
        W = np.zeros((self.S, self.S), np.float32)
        O = np.zeros((C, self.S), np.float32)
        lang = word >= 0
        T[word[lang], frm[lang], to[lang]] = 1
        wild = word == -1
        W[frm[wild], to[wild]] = 1
        O[label, to] = 1
        return T, W, O

    def enable_crf(self, transitions=None):
        """BASELINE config 4 (onehot + fused Viterbi): the composition SURVEY.md 8a-note defines --
        scores + two zero columns -> clamp column C'-3 -> CRF._viterbi_decode -> C'-3 -> o_idx."""
        self.use_crf = True
        self.crf_transitions = None if transitions is None else _lib.f32(transitions)
        self.invalidate()
        return self

    def _build_handle(self):
        a = self.args
        if a.local_loss_func != 'CE1':
            raise NotImplementedError('only CE1 is reachable from main.py (:127)')
        kw = dict(P=self._P(), nl=a.update_nonlinear, semiring='max' if a.train_mode == 'max' else 'sum',
                  threshold=a.threshold, o_idx=self.o_idx, use_crf=self.use_crf,
                  crf_trans=self.crf_transitions, device=self.device_index)
        if getattr(self, 'edges', None) is not None:
            word, frm, to, label = self.edges
            return _lib.create_onehot_ifst_from_edges(self.V, self.S, self.C, word, frm, to, label,
                                                      self.h0, self.hT, **kw)
        return _lib.create_onehot_ifst(self.language_tensor, self.wildcard_mat, self.output_mat, self.h0,
                                       self.hT, **kw)

    def state_dict(self):
        if getattr(self, 'edges', None) is not None:
            T, W, O = self._dense()
            return {'h0': self.h0, 'hT': self.hT, 'language_tensor': T, 'wildcard_mat': W, 'output_mat': O,
                    'output_wildcard_vector': np.zeros(self.S, np.float32)}
        return {'h0': self.h0, 'hT': self.hT, 'language_tensor': self.language_tensor,
                'wildcard_mat': self.wildcard_mat, 'output_mat': self.output_mat,
                'output_wildcard_vector': self.output_wildcard_vector}


class FARNN_S_O(_OnehotBase):
    def __init__(self, language_tensor=None, wildcard_tensor=None, wildcard_wildcard_mat=None,
                 final_vector=None, start_vector=None, priority_mat=None, args=None, o_idx=0,
                 is_cuda=False):
        C, S, _ = wildcard_tensor.shape
        super().__init__(args, o_idx, C, priority_mat, is_cuda)
        self.S = S
        self.h0 = _noisy(start_vector, self.amp)
        self.hT = _noisy(final_vector, self.amp)
        self.language_tensor = _noisy(language_tensor, self.amp)
        self.wildcard_tensor = _noisy(wildcard_tensor, self.amp)
        self.wildcard_wildcard_mat = _lib.f32(wildcard_wildcard_mat)

    @classmethod
    def from_automaton(cls, automata, word2idx, slot2idx, priority_mat=None, args=None, o_idx=0,
                       dataset='MITR-BIO'):
        """Built without the dense [V,C,S,S] host tensor (2.5 GB at ATIS size): see FARNN_S_O_I_S.from_automaton."""
        from ..wfa.fsa_to_tensor import dfa_to_edges_slot_new_wildcard
        if args.rand_constant:
            raise ValueError('from_automaton needs rand_constant == 0')
        word, frm, to, label, fin, sta, _ = dfa_to_edges_slot_new_wildcard(automata, word2idx, slot2idx, dataset)
        self = cls.__new__(cls)
        _OnehotBase.__init__(self, args, o_idx, len(slot2idx) + 1, priority_mat, False)
        self.S, self.V = len(automata['states']), len(word2idx)
        self.h0, self.hT = _noisy(sta, 0), _noisy(fin, 0)
        self.edges = (word, frm, to, label)
        return self

    def _build_handle(self):
        a = self.args
        if a.local_loss_func != 'CE1':
            raise NotImplementedError('only CE1 is reachable from main.py (:127)')
        if getattr(self, 'edges', None) is not None:
            word, frm, to, label = self.edges
            return _lib.create_onehot_fst4_from_edges(
                self.V, self.S, self.C, word, frm, to, label, self.h0, self.hT, P=self._P(),
                semiring='max' if a.train_mode == 'max' else 'sum', threshold=a.threshold,
                o_idx=self.o_idx, device=self.device_index)
        return _lib.create_onehot_fst4(
            self.language_tensor, self.wildcard_tensor, self.h0, self.hT, P=self._P(),
            semiring='max' if a.train_mode == 'max' else 'sum', threshold=a.threshold,
            o_idx=self.o_idx, device=self.device_index)

    def state_dict(self):
        return {'h0': self.h0, 'hT': self.hT, 'language_tensor': self.language_tensor,
                'wildcard_tensor': self.wildcard_tensor,
                'wildcard_wildcard_mat': self.wildcard_wildcard_mat}


class FARNN_S_O_I(_OnehotBase):
    def __init__(self, language_tensor=None, output_tensor=None, wildcard_mat=None,
                 output_wildcard_mat=None, final_vector=None, start_vector=None, priority_mat=None,
                 args=None, o_idx=0, is_cuda=False):
        C, S, _ = output_tensor.shape
        super().__init__(args, o_idx, C, priority_mat, is_cuda)
        self.S = S
        self.h0 = _noisy(start_vector, self.amp)
        self.hT = _noisy(final_vector, self.amp)
        self.language_tensor = _noisy(language_tensor, self.amp)
        self.wildcard_mat = _noisy(wildcard_mat, self.amp)
        self.output_tensor = _lib.f32(output_tensor)
        self.output_wildcard_mat = None if output_wildcard_mat is None else _lib.f32(output_wildcard_mat)

    @classmethod
    def from_automaton(cls, automata, word2idx, slot2idx, priority_mat=None, args=None, o_idx=0,
                       dataset='MITR-BIO'):
        """Built without the dense host tensors: see FARNN_S_O_I_S.from_automaton."""
        from ..wfa.fsa_to_tensor import dfa_to_edges_slot_independent_wildcard
        if args.rand_constant:
            raise ValueError('from_automaton needs rand_constant == 0')
        word, frm, to, label, fin, sta, _ = dfa_to_edges_slot_independent_wildcard(automata, word2idx, slot2idx,
                                                                                   dataset)
        self = cls.__new__(cls)
        _OnehotBase.__init__(self, args, o_idx, len(slot2idx) + 1, priority_mat, False)
        self.S, self.V = len(automata['states']), len(word2idx)
        self.h0, self.hT = _noisy(sta, 0), _noisy(fin, 0)
        self.edges = (word, frm, to, label)
        return self

    def _build_handle(self):
        a = self.args
        if a.local_loss_func != 'CE1':
            raise NotImplementedError('only CE1 is reachable from main.py (:127)')
        if getattr(self, 'edges', None) is not None:
            word, frm, to, label = self.edges
            return _lib.create_onehot_ind1_from_edges(
                self.V, self.S, self.C, word, frm, to, label, self.h0, self.hT, P=self._P(),
                semiring='max' if a.train_mode == 'max' else 'sum', mask_by_output=(a.independent == 2),
                threshold=a.threshold, o_idx=self.o_idx, device=self.device_index)
        return _lib.create_onehot_ind1(
            self.language_tensor, self.wildcard_mat, self.output_tensor, self.h0, self.hT, P=self._P(),
            semiring='max' if a.train_mode == 'max' else 'sum',
            mask_by_output=(a.independent == 2), threshold=a.threshold, o_idx=self.o_idx,
            device=self.device_index)

    def state_dict(self):
        return {'h0': self.h0, 'hT': self.hT, 'language_tensor': self.language_tensor,
                'wildcard_mat': self.wildcard_mat, 'output_tensor': self.output_tensor}

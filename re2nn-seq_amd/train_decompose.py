"""``--method decompose`` driver (reference src_seq/train_decompose.py:17-153): data,
decomposed-automaton pickle -> factors -> FARNN_S_D_W_I_S (``--independent 2``, what the shipped
example configs use) or FARNN_S_D_W_I (``--independent 1``), INIT evaluation, `.res` record.
``--independent 0`` (FARNN_S_D_W) is wired like the reference's (:72-90) but, like there, cannot be
reached from the CLI: main.py:127 forces local_loss_func='CE1' and :69-70 asserts independent != 0
under CE1.  The epoch loop (:155-220) runs on the HIP training step (csrc/train.hip.h through
train_onehot.train_epochs) for the decomposed i-FST (``--independent 2``) with farnn 0/1/2, with or without the CRF,
CE + CRF NLL losses; the KD / PR teacher terms and the other ``--independent`` settings are evaluated with --epoch 0."""
from .farnn.model_decompose import FARNN_S_D_W
from .farnn.model_decompose_independent import FARNN_S_D_W_I
from .farnn.model_decompose_single import FARNN_S_D_W_I_S
from .init_params import (get_init_params_seq, get_init_params_seq_independent,
                          get_init_params_seq_independent_single)
from .train_onehot import init_evaluation, prepare_slot_data
from .utils import Logger, set_seed


def train_slot_decompose(args, data_dir='../data/', model_dir='../model_seq/'):
    logger = Logger()
    set_seed(args.seed)
    t2i, i2t, s2i, i2s, splits = prepare_slot_data(args, data_dir)
    print('Train Samples: ', len(splits['train']))
    if args.local_loss_func == 'CE1':
        assert args.independent != 0               # (:69-70)
    if args.independent == 0:                      # 4th-order tensor model (:72-90)
        (V_embed_extend, C_embed, S1, S2, pretrain_embed_extend, wildcard_tensor, wildcard_wildcard_tensor,
         final_vector, start_vector, priority_mat, C_wildcard, S1_wildcard, S2_wildcard) = \
            get_init_params_seq(args, s2i, data_dir=data_dir)
        model = FARNN_S_D_W(V=V_embed_extend, C=C_embed, S1=S1, S2=S2, C_wildcard=C_wildcard,
                            S1_wildcard=S1_wildcard, S2_wildcard=S2_wildcard,
                            wildcard_wildcard=wildcard_wildcard_tensor, final_vector=final_vector,
                            start_vector=start_vector, pretrained_word_embed=pretrain_embed_extend,
                            priority_mat=priority_mat, args=args, o_idx=s2i['o'])
        return init_evaluation(model, splits, args, s2i, i2s, logger, model_dir)
    if args.independent != 2:                      # two 3rd-order tensors (:112-131)
        (V_embed_extend, S1, S2, pretrain_embed_extend, wildcard_mat, wildcard_output,
         final_vector, start_vector, priority_mat, C_output, S1_output, S2_output) = \
            get_init_params_seq_independent(args, s2i, t2i, data_dir=data_dir)
        model = FARNN_S_D_W_I(V=V_embed_extend, S1=S1, S2=S2, C_output=C_output, S1_output=S1_output,
                              S2_output=S2_output, wildcard_mat=wildcard_mat,
                              wildcard_output=wildcard_output, final_vector=final_vector,
                              start_vector=start_vector, pretrained_word_embed=pretrain_embed_extend,
                              priority_mat=priority_mat, args=args, o_idx=s2i['o'])
        return init_evaluation(model, splits, args, s2i, i2s, logger, model_dir)
    t2i_nopad = t2i                                # init_params appends the pad row itself
    (V_embed_extend, S1, S2, pretrain_embed_extend, wildcard_mat, wildcard_output_vector,
     final_vector, start_vector, priority_mat, C_output_mat, _) = \
        get_init_params_seq_independent_single(args, s2i, t2i_nopad, data_dir=data_dir)
    model = FARNN_S_D_W_I_S(V=V_embed_extend, S1=S1, S2=S2, C_output_mat=C_output_mat,
                            wildcard_mat=wildcard_mat, wildcard_output_vector=wildcard_output_vector,
                            final_vector=final_vector, start_vector=start_vector,
                            pretrained_word_embed=pretrain_embed_extend, priority_mat=priority_mat,
                            args=args, o_idx=s2i['o'])
    return init_evaluation(model, splits, args, s2i, i2s, logger, model_dir)

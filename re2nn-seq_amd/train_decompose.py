"""``--method decompose`` driver (reference src_seq/train_decompose.py:17-153): data,
decomposed-automaton pickle -> factors -> FARNN_S_D_W_I_S, INIT evaluation, `.res` record.
Only ``--independent 2`` (the i-FST, what the shipped example configs use) is on the tagging
path built here; the epoch loop (:155-220) is out of scope like in train_onehot.py."""
from .farnn.model_decompose_single import FARNN_S_D_W_I_S
from .init_params import get_init_params_seq_independent_single
from .train_onehot import init_evaluation, prepare_slot_data
from .utils import Logger, set_seed


def train_slot_decompose(args, data_dir='../data/', model_dir='../model_seq/'):
    logger = Logger()
    set_seed(args.seed)
    t2i, i2t, s2i, i2s, splits = prepare_slot_data(args, data_dir)
    print('Train Samples: ', len(splits['train']))
    if args.local_loss_func == 'CE1':
        assert args.independent != 0               # (:69-70)
    if args.independent != 2:
        raise NotImplementedError(
            '--independent {} decomposed variants (FARNN_S_D_W / FARNN_S_D_W_I) are listed as '
            '"next" in SURVEY.md 8f; use --independent 2'.format(args.independent))
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

"""Command line of the tagger: the reference's ``src_seq/main.py`` flag set, sanity checks and
dispatch (ref :14-100 flags, :108-123 ``--args_path`` reload, :126-186 asserts, :188-199
dispatch), so existing launch scripts keep working:

    python -m re2nn_seq_amd.main --dataset ATIS-BIO --method onehot --independent 2 \
        --automata_path ../data/ATIS-BIO/automata/<name>.ID2 --normalize_automata none \
        --rand_constant 0 --bz 256 --seq_max_len 64 --epoch 0 --train_portion 0

Two extra, optional flags exist for running outside the reference's directory layout:
``--data_dir`` (default ../data/) and ``--model_dir`` (default ../model_seq/).
"""
import argparse
import pickle
import sys

# (flag, type, default, help) -- one row per reference flag, same order as main.py:19-98
FLAGS = [
    ('dataset', str, 'SNIPS-BIO', 'dataset dir'),
    ('seq_max_len', int, 30, 'Max seq length'),
    ('bz', int, 500, 'batch size'),
    ('embed_dim', int, 100, 'embed dim'),
    ('embed_type', str, 'glove', 'embedding type should be in [glove, fasttext]'),
    ('epoch', int, 20, 'number of training epochs (0 = evaluate the initial model only)'),
    ('train_portion', float, 1.0, 'train portion'),
    ('automata_path', str, '../data/MITR-toy/automata/automata.dict', 'automata path'),
    ('seed', int, 0, 'random seed'),
    ('run', str, 'test', 'run string'),
    ('random_embed', int, 0, '0 false 1 true'),
    ('optimizer', str, 'ADAM', 'optimizer'),
    ('lr', float, 0.0001, 'learning rate of optimizer'),
    ('train_mode', str, 'sum', 'global train mode, should be in [max, sum]'),
    ('local_loss_func', str, 'CE1', 'loss function in local mode'),
    ('rand_constant', float, 1e-5, 'random noise'),
    ('threshold', float, 0.5, 'clamp of the wildcard (oo) score when decoding'),
    ('margin', float, 0.3, 'margin of the hinge loss option'),
    ('select_level', str, 'entity-level', 'entity-level or token-level'),
    ('method', str, 'onehot', 'method should be in [onehot, decompose, baseline]'),
    ('data_type', str, 'all', 'data type we use, should be in [all, re, n_re]'),
    # baselines
    ('train_word_embed', int, 0, 'if we train word embed or not'),
    ('rnn_hidden_dim', int, 100, 'rnn / farnn_random hidden dim'),
    ('rnn', str, 'RNN', 'should be in RNN, LSTM, GRU'),
    ('bidirection', int, 0, '1 means bidirectional'),
    ('marryup_type', str, 'none', 'marryup type, [input, output, all, kd, pr]'),
    ('re_tag_dim', int, 20, 're tag embedding dim for marryup methods'),
    ('c1_kdpr', float, 1, 'regularization param for PR / temperature in KD'),
    ('c2_kdpr', float, 1, 'balancing weight for KD PR loss and original loss'),
    ('c3_pr', float, 1, 'annealing speed for pr'),
    # decomposed
    ('normalize_automata', str, 'l2-rank', 'normalisation of the decomposed factors [none, l1, l2, l1-rank, l2-rank]'),
    ('train_V_embed', int, 0, '0 means do not train V_embed'),
    ('beta', float, 1.0, 'interpolation weight for word embedding and rule embedding'),
    ('rank', int, 150, 'rank of decomposed tensor'),
    ('rank_wildcard', int, 50, 'rank of wildcard decomposed tensor'),
    ('additional_nonlinear', str, 'none', 'additional nonlinear for word embedding to rule dim'),
    ('additional_states', int, 0, 'additional states with very small random values'),
    ('use_priority', int, 0, '0, or 1, 1 means use priority'),
    ('train_wildcard', int, 0, 'if we train wildcard tensor CxSxS'),
    ('train_wildcard_wildcard', int, 0, 'if we train wildcard_wildcard matrix SxS'),
    ('train_c_output', int, 1, 'if we train C related params in single'),
    ('train_h0', int, 0, 'if we train h0'),
    ('train_hT', int, 0, 'if we train hT'),
    ('train_beta', int, 0, 'if we train beta'),
    ('random', int, 0, 'if we use random initialization'),
    ('random_pad_func', str, 'uniform', 'padding function: normal, uniform, xavier'),
    ('save_model', int, 0, 'if we save model'),
    ('independent', int, 0, '0: FST 4-D tensor, 1: two 3-D tensors, 2: i-FST'),
    ('use_unlabel', int, 0, 'if we use unlabel data'),
    # FA-GRU
    ('farnn', int, 0, '0 for rnn, 1 for only update, 2 for update + reset'),
    ('xavier', int, 0, 'xavier init of the gate parameters'),
    ('bias_init', float, 5, 'initial gate bias'),
    ('sigmoid_exponent', int, 5, 'sigmoidal function exponent'),
    ('use_crf', int, 0, 'if we use crf'),
    ('update_nonlinear', str, 'none', 'nonlinearity applied to the state update'),
    # save / load
    ('args_path', str, 'none', 'arguments path, if is not none, load and run'),
    # BERT
    ('bert_finetune', int, 0, 'if we finetune bert'),
    ('use_bert', int, 0, 'if we use bert'),
    ('warm_up', int, 0, 'if we use warm up'),
    ('bert_lr_down_factor', float, 1, 'the down factor for the bert lr'),
    ('bert_init_embed', str, 'aggregate', 'embed used to initializing G'),
]
EXTRA_FLAGS = [
    ('data_dir', str, '../data/', '[extra] root of the <dataset>/dataset.pkl tree'),
    ('model_dir', str, '../model_seq/', '[extra] where .res files are written'),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, typ, default, help_ in FLAGS + EXTRA_FLAGS:
        parser.add_argument('--' + name, type=typ, default=default, help=help_)
    return parser


def parse_args(argv=None):
    parser = build_parser()
    return parser.parse_args(argv), parser


class _ResUnpickler(pickle.Unpickler):
    """`.res` files pickle reference classes (src_seq.tools.printer.Best_Model_Recorder,
    src_seq.utils.Logger); resolve them to this package's counterparts."""

    def find_class(self, module, name):
        if module.startswith('src_seq'):
            from . import utils
            from .tools import printer
            for mod in (printer, utils):
                if hasattr(mod, name):
                    return getattr(mod, name)
            return type(name, (), {})
        return super().find_class(module, name)


def load_res(path):
    with open(path, 'rb') as f:
        return _ResUnpickler(f).load()


def merge_saved_args(args, path):
    """ref :108-123: saved hyper-parameters override the command line; run is forced."""
    loaded = load_res(path)['args'].__dict__
    merged = dict(args.__dict__)
    for k in loaded:
        if k not in merged:
            print(k)
    for k in merged:
        if k in loaded and k not in ('data_dir', 'model_dir'):
            merged[k] = loaded[k]
        elif k not in loaded:
            print(k)
    print(merged)
    out = argparse.Namespace(**merged)
    out.run = 'final_222'
    return out


def check_args(args, parser):
    """ref :126-186, same conditions in the same order."""
    assert args.train_mode in ['max', 'sum']
    assert args.local_loss_func in ['CE1']
    assert args.update_nonlinear in ['none', 'relu', 'tanh', 'relutanh']
    assert args.rnn in ['LSTM', 'RNN', 'GRU']
    assert args.method in ['decompose', 'onehot', 'baseline']
    assert args.normalize_automata in ['none', 'l1', 'l2', 'l1-rank', 'l2-rank']
    assert args.additional_nonlinear in ['none', 'relu', 'tanh', 'sigmoid', 'relutanh']
    assert args.select_level in ['entity-level', 'token-level']
    assert args.rank in [30, 100, 150, 200, 250, 300, 350]
    assert args.rank_wildcard in [20, 30, 50, 70, 100, 150]
    assert args.random_pad_func in ['normal', 'xavier', 'uniform']
    assert args.seed in [0, 1, 2, 3, 4, 5]
    assert args.data_type in ['all', 're', 'n_re']
    assert args.independent in [0, 1, 2]
    if args.bert_finetune == 1:
        assert args.bert_lr_down_factor >= 5
    if args.train_portion == 0:
        assert args.epoch == 0
    if args.normalize_automata != 'none':
        assert args.method == 'decompose'
    if args.select_level == 'entity-level':
        assert 'BIO' in args.dataset
    if args.use_crf == 1:
        assert args.local_loss_func in ['CE', 'CE1']
    if args.random == 1:
        assert args.method != 'baseline'
    if args.method == 'decompose':
        assert args.marryup_type in ['none', 'kd', 'pr']
    if args.method == 'baseline':
        assert args.marryup_type in ['none', 'input', 'output', 'all', 'pr', 'kd']
        if args.marryup_type == 'kd':
            assert args.c3_pr == parser.get_default('c3_pr')
            assert args.c1_kdpr >= 1.0
        elif args.marryup_type == 'pr':
            assert args.c1_kdpr >= 1.0
    if args.method == 'onehot':
        assert args.rand_constant == 0
    assert args.embed_type in ['glove', 'fasttext']
    assert args.dataset in ['ATIS-BIO', 'ATIS-ZH-BIO', 'SNIPS-BIO']
    if args.dataset == 'ATIS-ZH-BIO':
        assert args.embed_type == 'fasttext'
    if not bool(args.use_bert):
        assert args.warm_up == 0
        assert args.bert_finetune == 0
        assert args.bert_lr_down_factor == 1


def dispatch(args):
    """ref :188-199.  The BiRNN/BERT baselines are comparison systems, not the FA-RNN tagging
    path."""
    if args.method == 'onehot':
        from .train_onehot import train_slot_onehot
        return train_slot_onehot(args, data_dir=args.data_dir, model_dir=args.model_dir)
    if args.method == 'decompose':
        if args.use_bert:
            raise NotImplementedError('the BERT front-end is out of scope (SURVEY.md 2, row 23)')
        from .train_decompose import train_slot_decompose
        return train_slot_decompose(args, data_dir=args.data_dir, model_dir=args.model_dir)
    raise NotImplementedError('--method baseline (BiRNN / MarryUp baselines) is out of scope '
                              '(SURVEY.md 2, row 22)')


def main(argv=None):
    args, parser = parse_args(argv)
    if args.args_path != 'none':
        args = merge_saved_args(args, args.args_path)
    check_args(args, parser)
    return dispatch(args)


if __name__ == '__main__':
    main(sys.argv[1:])

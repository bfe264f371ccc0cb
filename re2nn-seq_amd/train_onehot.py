"""``--method onehot`` driver (reference src_seq/train_onehot.py:20-154: data, automaton ->
tensors -> model, INIT evaluation on train/dev/test, `.res` record).  The epoch loop of the
reference (:156-206, backward pass + Adam) is outside the forward tagging path: with
``--epoch 0`` (what ``--train_portion 0`` requires, main.py:147-148) this driver is complete;
a positive epoch count is refused."""
from .create_logic_mat_bias import create_mat_priority_MITR
from .data import SlotBatchDataset, iter_batches, load_slot_dataset
from .RE import build_onehot_model
from .tools.printer import Best_Model_Recorder, print_and_log_results
from .tools.saver import save_model_and_log
from .utils import Logger, len_stats, load_pkl, pad_dataset_1, set_seed
from .val import val_onehot


def prepare_slot_data(args, data_dir='../data/'):
    """Load dataset.pkl, append '<pad>' as the last vocabulary id and pad every split
    (ref :36-60)."""
    dset = load_slot_dataset(args.dataset, data_dir)
    t2i, i2t, s2i, i2s = dset['t2i'], dset['i2t'], dset['s2i'], dset['i2s']
    for name in ('train', 'dev', 'test'):
        len_stats(dset['query_' + name])
    i2t[len(i2t)] = '<pad>'
    t2i['<pad>'] = len(i2t) - 1
    splits = {}
    for name in ('train', 'dev', 'test'):
        q, _, lens = pad_dataset_1(dset['query_' + name], args.seq_max_len, t2i['<pad>'])
        s, _, _ = pad_dataset_1(dset['intent_' + name], args.seq_max_len, s2i['o'])
        portion = 1 if name == 'test' else args.train_portion
        splits[name] = SlotBatchDataset(q, lens, s, args, s2i, portion=portion, dset=name, data_dir=data_dir)
    return t2i, i2t, s2i, i2s, splits


def init_evaluation(model, splits, args, s2i, i2s, logger, model_dir='../model_seq/'):
    """INIT eval on the three splits + best-model record + `.res` (ref :142-154, :208)."""
    results, stats = {}, {}
    for name, mode in (('train', 'TRAIN'), ('dev', 'DEV'), ('test', 'TEST')):
        st = {}
        results[name] = val_onehot(iter_batches(splits[name], args.bz), model, args, s2i['o'], i2s, stats=st)
        print_and_log_results(logger, results[name], 'INIT', mode)
        stats[name] = st
        info = 'THROUGHPUT | {} | {} tokens in {:.4f} s = {:.1f} tokens/s'.format(
            mode, st['tokens'], st['seconds'], st['tokens_per_s'])
        print(info)
        logger.add(info)
    recorder = Best_Model_Recorder(selector='f', level=args.select_level,
                                   init_results_train=results['train'], init_results_dev=results['dev'],
                                   init_results_test=results['test'],
                                   save_model=bool(getattr(args, 'save_model', 0)))
    if args.epoch > 0:
        raise NotImplementedError(
            'training epochs (backward pass, optimizer) are outside the forward tagging path this '
            'package accelerates; run with --epoch 0 (see DESIGN.md, out of scope)')
    path = save_model_and_log(logger, recorder, args, model_dir=model_dir)
    return results, stats, path


def train_slot_onehot(args, data_dir='../data/', model_dir='../model_seq/'):
    logger = Logger()
    set_seed(args.seed)
    t2i, i2t, s2i, i2s, splits = prepare_slot_data(args, data_dir)
    print('Train Samples: ', len(splits['train']))
    automata = load_pkl(args.automata_path)
    if 'automata' in automata:                     # (:71-72)
        automata = automata['automata']
    print("AUTOMATA STATES NUM: {}".format(len(automata['states'])))
    # the onehot driver always uses the identity priority matrix (:68)
    model = build_onehot_model(args, automata, t2i, s2i, create_mat_priority_MITR(s2i))
    if getattr(args, 'use_crf', 0) and hasattr(model, 'enable_crf'):
        model.enable_crf()                         # BASELINE config 4
    return init_evaluation(model, splits, args, s2i, i2s, logger, model_dir)

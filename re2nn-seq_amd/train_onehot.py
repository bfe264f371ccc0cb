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
        if not hasattr(model, 'enable_training'):
            raise NotImplementedError(
                'training epochs are implemented for the decomposed i-FST (--method decompose --independent 2, '
                'farnn 0, no CRF: DESIGN.md row f3); run the other models with --epoch 0')
        train_epochs(model, splits, args, s2i, i2s, logger, recorder, stats)
    path = save_model_and_log(logger, recorder, args, model_dir=model_dir)
    return results, stats, path


def train_epochs(model, splits, args, s2i, i2s, logger, recorder, stats):
    """The epoch loop of the reference (train_decompose.py:161-221): forward_local(train=True), loss.backward(),
    optimizer.step() per batch, then the three evaluations and the best-model record."""
    import time

    import torch
    from .metrics.metrics import eval_seq_token, get_ner_fmeasure

    model.enable_training()                       # raises for the configurations the HIP training step does not cover
    params = list(model.parameters())
    if args.optimizer == 'SGD':
        optimizer = torch.optim.SGD(params, lr=args.lr, weight_decay=0)
    else:
        optimizer = torch.optim.Adam(params, lr=args.lr, weight_decay=0)
    print('ALL TRAINABLE PARAMETERS: {}'.format(sum(p.numel() for p in params)))
    for epoch in range(1, args.epoch + 1):
        model.train()
        preds, trues, avg_loss, n_tok = [], [], 0.0, 0
        t0 = time.perf_counter()
        for batch in iter_batches(splits['train'], args.bz):
            optimizer.zero_grad()
            loss, pred, true = model.forward_local(batch['x'], batch['s'], batch['l'], train=True)
            loss.backward()
            optimizer.step()
            avg_loss += float(loss.detach())
            preds.append(pred.cpu())
            trues.append(true.cpu())
            n_tok += int(batch['l'].sum())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        avg_loss /= max(len(splits['train']), 1)
        info = '{} Epoch: {} | LOSS: {}'.format('TRAIN', epoch, avg_loss)
        print(info)
        logger.add(info)
        info = 'THROUGHPUT | TRAIN-STEP | {} tokens in {:.4f} s = {:.1f} tokens/s'.format(n_tok, el, n_tok / el)
        print(info)
        logger.add(info)
        stats.setdefault('train_step', []).append({'tokens': n_tok, 'seconds': el, 'tokens_per_s': n_tok / el})
        all_pred, all_true = torch.cat(preds), torch.cat(trues)
        res_train = {'token-level': list(eval_seq_token(seq_label_pred=all_pred, seq_label_true=all_true, o_idx=s2i['o'])),
                     'entity-level': list(get_ner_fmeasure(golden_lists=all_true, predict_lists=all_pred, i2s=i2s))}
        print_and_log_results(logger, res_train, epoch, 'TRAIN')
        res_dev = val_onehot(iter_batches(splits['dev'], args.bz), model, args, s2i['o'], i2s)
        print_and_log_results(logger, res_dev, epoch, 'DEV')
        res_test = val_onehot(iter_batches(splits['test'], args.bz), model, args, s2i['o'], i2s)
        print_and_log_results(logger, res_test, epoch, 'TEST')
        recorder.update_and_record(res_train, res_dev, res_test, model.state_dict())


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

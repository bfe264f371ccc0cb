"""Result printing and best-model bookkeeping (reference src_seq/tools/printer.py).  Kept
attribute-compatible because instances are pickled into the `.res` files that `--args_path`
reads back."""
from copy import deepcopy


def print_and_log_results(logger, results, epoch, mode):
    assert mode in ['TRAIN', 'DEV', 'TEST', 'DEV_RE', 'DEV_NO_RE']
    for level, name in (('token-level', 'TOKEN'), ('entity-level', 'ENTITY')):
        acc, p, r, f = results[level][:4]
        info = '{} | {} EPOCH {} |  ACC: {}, P: {}, R:{}, F1: {}'.format(name, mode, epoch, acc, p, r, f)
        print(info)
        logger.add(info)
    info = str(results['entity-level'][4])
    print(info)
    logger.add(info)


class Best_Model_Recorder:
    def __init__(self, selector='f', level='token-level', init_results_train=None,
                 init_results_dev=None, init_results_test=None, save_model=False):
        pool = ['p', 'r', 'f']
        assert selector in pool and level in ['token-level', 'entity-level']
        self.best_dev_results = init_results_dev
        self.best_dev_train_results = init_results_train
        self.best_dev_test_results = init_results_test
        self.selector = pool.index(selector) + 1
        self.level = level
        self.best_selector = self.best_dev_results[self.level][self.selector]
        self.best_model_state_dict = None
        self.save_model = save_model

    def update_and_record(self, results_train, results_dev, results_test, model_state_dict):
        value = results_dev[self.level][self.selector]
        if value > self.best_selector:
            self.best_selector = value
            self.best_dev_results, self.best_dev_test_results = results_dev, results_test
            self.best_dev_train_results = results_train
            if self.save_model:
                self.best_model_state_dict = deepcopy(model_state_dict)

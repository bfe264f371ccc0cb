"""`.res` writer (reference src_seq/tools/saver.py:4-18): {'args', 'res', 'logger'} pickled to
<model_dir>/<run>/<timestamp>.res (model_dir defaults to the reference's ../model_seq/)."""
import os
import pickle

from ..utils import create_datetime_str, mkdir


def save_model_and_log(logger, result, args, model_dir='../model_seq/'):
    run_dir = os.path.join(model_dir, str(args.run))
    mkdir(run_dir)
    path = os.path.join(run_dir, create_datetime_str() + '.res')
    print('Saving Args and Results at: {}'.format(path))
    with open(path, 'wb') as f:
        pickle.dump({'args': args, 'res': result, 'logger': logger}, f)
    return path

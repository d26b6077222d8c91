"""nerv.utils: the object (de)serialisation helpers the slot-file drivers use (extract_slots.py:15,75;
rollout_clevrer_slots.py:13,100).  `.pkl` = pickle, `.npy` = numpy, `.json` = json -- chosen by suffix."""
import json
import os
import pickle

import numpy as np


def mkdir_or_exist(path):
    if path:
        os.makedirs(path, exist_ok=True)


def dump_obj(obj, path, **kwargs):
    mkdir_or_exist(os.path.dirname(os.path.abspath(path)))
    ext = os.path.splitext(path)[1].lower()
    if ext in ('.pkl', '.pickle'):
        with open(path, 'wb') as f:
            pickle.dump(obj, f, **kwargs)
    elif ext == '.npy':
        np.save(path, obj)
    elif ext == '.json':
        with open(path, 'w') as f:
            json.dump(obj, f, **kwargs)
    else:
        raise ValueError(f'dump_obj: unsupported suffix {ext!r}')


def load_obj(path, **kwargs):
    ext = os.path.splitext(path)[1].lower()
    if ext in ('.pkl', '.pickle'):
        with open(path, 'rb') as f:
            return pickle.load(f, **kwargs)
    if ext == '.npy':
        return np.load(path, **kwargs)
    if ext == '.json':
        with open(path) as f:
            return json.load(f, **kwargs)
    raise ValueError(f'load_obj: unsupported suffix {ext!r}')


def strip_suffix(path):
    return os.path.splitext(path)[0]

"""Dataset wire format and loader conventions of the reference.

Writer: ``create_dataset.py:75-79`` -> ``{"data": [ndarray(200,12) f64, ...], "stiffness": [float, ...]}`` pickled.
Reader: ``training_cross_validate.py:20-33`` (several pickles concatenated) and ``functions/utils.py:5-43`` (index split,
per-channel mean/std of the training part over axes (0,1), keepdims).
"""
import pickle

import numpy as np


def save_dataset(path, data, stiffness):
    data = [np.asarray(x, dtype=np.float64) for x in data]
    assert all(x.ndim == 2 for x in data) and len(data) == len(stiffness)
    with open(path, "wb") as f:
        pickle.dump({"data": data, "stiffness": [float(k) for k in stiffness]}, f)


def load_datasets(paths):
    """concatenate pickles the way training_cross_validate.py:20-33 does"""
    out = {"data": [], "stiffness": []}
    for p in paths:
        with open(p, "rb") as f:
            d = pickle.load(f)
        out["data"].extend(d["data"])
        out["stiffness"].extend(d["stiffness"])
    return out


def split_and_stats(dataset, train_idx, val_idx):
    """functions/utils.py:7-41: returns (train_x, train_y, val_x, val_y, mean, std)"""
    x, y = np.array(dataset["data"]), np.array(dataset["stiffness"])
    train_x, train_y, val_x, val_y = x[list(train_idx)], y[list(train_idx)], x[list(val_idx)], y[list(val_idx)]
    mean = np.mean(train_x, axis=(0, 1), keepdims=True)
    std = np.std(train_x, axis=(0, 1), keepdims=True)
    return train_x, train_y, val_x, val_y, mean, std

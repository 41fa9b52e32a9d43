"""helpers — the one helper on the hot path: roundoff (reference code/helpers.py:39-46)."""
import numpy as np


def roundoff(Y):
    y_ = []
    for y in Y:
        if y >= 0.5:
            y_.append([1])
        else:
            y_.append([0])
    return np.stack(y_)


def one_hot(Y, n_classes):
    """reference code/helpers.py:32-36"""
    y_ = np.zeros((len(Y), n_classes))
    y_[np.arange(len(Y)), Y] = 1
    return y_

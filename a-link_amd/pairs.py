"""pairs — the pair-enumeration workload of one A-LINK iteration.

createMiniBatch restates reference code/readDFW.py:222-244 (identical in readMTP.py:123-135): all
(plain_i x disguised_j) pairs, then all (disguised_i x disguised_j) pairs, label 1 iff i == j.
createMiniBatchIndices is the de-duplicated form: the same pair list as
indices into the unique-image array, so each image is embedded once and pairs are gathered on device.
"""
import numpy as np


def createMiniBatch(X_plain, X_dig):
    X_left, X_right, Y = [], [], []
    for i in range(len(X_plain)):
        for j in range(len(X_dig)):
            for x in X_plain[i]:
                for y in X_dig[j]:
                    X_left.append(x)
                    X_right.append(y)
                    Y.append([1] if i == j else [0])
    for i in range(len(X_dig)):
        for j in range(len(X_dig)):
            for x in X_dig[i]:
                for y in X_dig[j]:
                    X_left.append(x)
                    X_right.append(y)
                    Y.append([1] if i == j else [0])
    return [np.stack(X_left), np.stack(X_right)], np.stack(Y)


def createMiniBatchIndices(n_plain, n_dig):
    """n_plain[i], n_dig[i]: image counts per person.  Returns (li, ri, y): indices into the array
    `unique = concat(plain_0..plain_k, dig_0..dig_k)` reproducing createMiniBatch's order."""
    n_plain, n_dig = list(n_plain), list(n_dig)
    p_off = np.concatenate([[0], np.cumsum(n_plain)]).astype(np.int64)
    d_off = p_off[-1] + np.concatenate([[0], np.cumsum(n_dig)]).astype(np.int64)
    li, ri, y = [], [], []
    for i in range(len(n_plain)):
        for j in range(len(n_dig)):
            a = np.arange(p_off[i], p_off[i + 1])
            b = np.arange(d_off[j], d_off[j + 1])
            li.append(np.repeat(a, len(b)))
            ri.append(np.tile(b, len(a)))
            y.append(np.full(len(a) * len(b), 1 if i == j else 0))
    for i in range(len(n_dig)):
        for j in range(len(n_dig)):
            a = np.arange(d_off[i], d_off[i + 1])
            b = np.arange(d_off[j], d_off[j + 1])
            li.append(np.repeat(a, len(b)))
            ri.append(np.tile(b, len(a)))
            y.append(np.full(len(a) * len(b), 1 if i == j else 0))
    return (np.concatenate(li).astype(np.int32), np.concatenate(ri).astype(np.int32),
            np.concatenate(y).astype(np.int64).reshape(-1, 1))


def splitDisguiseData(X_dig, pre_ratio=0.5):
    """reference code/readDFW.py:212-219"""
    X_dig_pre, X_dig_post = [], []
    for i in range(len(X_dig)):
        splitPoint = int(X_dig[i].shape[0] * pre_ratio)
        X_dig_pre.append(X_dig[i][:splitPoint])
        X_dig_post.append(X_dig[i][splitPoint:])
    return (X_dig_pre, X_dig_post)

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


# ---- in-memory pair generators of the drivers (host Python, like the reference) -----------------------
def getNormalGenerator(X_data, batch_size, infinite=True):
    """readDFW.getNormalGenerator (code/readDFW.py:143-160): all (person i x person j) pairs in
    order, label 1 iff i == j, cut into batches of batch_size; the tail shorter than a batch is
    dropped at the end of a sweep."""
    while True:
        X_left, X_right, Y = [], [], []
        for i in range(len(X_data)):
            for j in range(len(X_data)):
                for x in X_data[i]:
                    for y in X_data[j]:
                        X_left.append(x)
                        X_right.append(y)
                        Y.append([1] if i == j else [0])
                        if len(Y) == batch_size:
                            yield [np.stack(X_left), np.stack(X_right)], np.stack(Y)
                            X_left, X_right, Y = [], [], []
        if not infinite:
            break


def getImposterGenerator(X_plain, X_imposter, batch_size, infinite=True):
    """readDFW.getImposterGenerator (code/readDFW.py:163-177): every plain image x every impostor image, label 0."""
    while True:
        X_left, X_right, Y = [], [], []
        for person in X_plain:
            for x in person:
                for imposter in X_imposter:
                    for y in imposter:
                        X_left.append(x)
                        X_right.append(y)
                        Y.append([0])
                        if len(Y) == batch_size:
                            yield [np.stack(X_left), np.stack(X_right)], np.stack(Y)
                            X_left, X_right, Y = [], [], []
        if not infinite:
            break


def _balanced(X, Y):
    """1:1 positives/negatives by np.random.choice without replacement (code/readDFW.py:189-199)."""
    Y_flat = np.stack([y[0] for y in Y])
    pos = np.where(Y_flat == 1)[0]
    neg = np.where(Y_flat == 0)[0]
    minSamp = np.minimum(len(pos), len(neg))
    if minSamp == 0:
        return None
    sel = np.concatenate((np.random.choice(pos, minSamp, replace=False), np.random.choice(neg, minSamp, replace=False)),
                         axis=0)
    return [X[0][sel], X[1][sel]], Y[sel]


def getGenerator(norGen, normImpGen, impGen, batch_size, type=0, val_ratio=0.2):
    """readDFW.getGenerator (code/readDFW.py:180-209).  The reference labels the third block with Y2
    again (`Y = concatenate((Y1, Y2, Y2))`, :185) — kept.  Ends (StopIteration) when a finite source
    generator is exhausted, where the Python-3 copy yields None (code/readDFW3.py)."""
    X_left, X_right, Y_send = [], [], []
    while True:
        try:
            X1, Y1 = next(norGen)
            X2, Y2 = next(normImpGen)
            X3, Y3 = next(impGen)
        except StopIteration:
            return
        Y = np.concatenate((Y1, Y2, Y2), axis=0)
        X = [np.concatenate((X1[0], X2[0], X3[0]), axis=0), np.concatenate((X1[1], X2[1], X3[1]), axis=0)]
        picked = _balanced(X, Y)
        if picked is None:
            continue
        X, Y = picked
        if len(Y_send) > 0:
            X_left = np.concatenate((X_left, X[0]), axis=0)
            X_right = np.concatenate((X_right, X[1]), axis=0)
            Y_send = np.concatenate((Y_send, Y), axis=0)
        else:
            X_left, X_right, Y_send = np.copy(X[0]), np.copy(X[1]), np.copy(Y)
        if len(Y_send) >= batch_size:
            yield ([X_left, X_right], Y_send)
            X_left, X_right, Y_send = [], [], []


def getGeneratorMTP(datGen, batch_size, resize_res=None, featurize=None):
    """readMTP.getGenerator (code/readMTP.py:80-113): balance, optionally resize (bilinear, on the
    device) and featurize each source batch, accumulate to batch_size."""
    from . import noise as _noise
    X_left, X_right, Y_send = [], [], []
    while True:
        try:
            X, Y = next(datGen)
        except StopIteration:
            return
        picked = _balanced(X, Y)
        if picked is None:
            continue
        X, Y = picked
        if resize_res:
            X = [np.asarray(_noise.resize_images(X[0], resize_res)), np.asarray(_noise.resize_images(X[1], resize_res))]
        if featurize:
            X = [featurize.process(X[0]), featurize.process(X[1])]
        if len(Y_send) > 0:
            X_left = np.concatenate((X_left, X[0]), axis=0)
            X_right = np.concatenate((X_right, X[1]), axis=0)
            Y_send = np.concatenate((Y_send, Y), axis=0)
        else:
            X_left, X_right, Y_send = np.copy(X[0]), np.copy(X[1]), np.copy(Y)
        if len(Y_send) >= batch_size:
            yield ([X_left, X_right], Y_send)
            X_left, X_right, Y_send = [], [], []

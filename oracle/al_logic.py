"""oracle/al_logic.py — NumPy restatement of the active-learning host arithmetic around the models.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/{uncertainty,bagging,minibatch}.npz,
which were produced by importing the reference's own functions (tests/golden/make_golden.py):
    _proba_uncertainty / _proba_margin / _proba_entropy      reference code/uncertainty.py:15-60
    uncertainty_sampling / margin_sampling / entropy_sampling code/uncertainty.py:133-217
    Bagging.predict                                           code/committee.py:13-20
    createMiniBatch / splitDisguiseData                       code/readDFW.py:212-244
The driver's selection rule is inline script code (code/ALINK_arc.py:167-198, code/ALINK.py:170-201)
and cannot be imported; select_queries restates it line by line and is pinned by hand-made
known-answer cases in tests/test_selection.py.
"""
import numpy as np
from scipy.stats import entropy


def proba_uncertainty(proba):
    return 1 - np.max(proba, axis=1)


def proba_margin(proba):
    if proba.shape[1] == 1:
        return np.zeros(shape=len(proba))
    part = np.partition(-proba, 1, axis=1)
    return -part[:, 0] + part[:, 1]


def proba_entropy(proba):
    return np.transpose(entropy(np.transpose(proba)))


def multi_argmax(values, n_instances=1):
    """modAL.utils.selection.multi_argmax (third-party, modAL 0.3.x; version unpinned by the
    reference): the n largest, order inside the n not contractual."""
    return np.argpartition(-values, n_instances - 1, axis=0)[:n_instances]


def bagging_predict(preds):
    predicted = np.sum(np.array(preds), axis=0) / len(preds)
    return np.array(predicted)


def create_minibatch(X_plain, X_dig):
    X_left, X_right, Y = [], [], []
    for i in range(len(X_plain)):
        for j in range(len(X_dig)):
            for x in X_plain[i]:
                for y in X_dig[j]:
                    X_left.append(x); X_right.append(y); Y.append([1] if i == j else [0])
    for i in range(len(X_dig)):
        for j in range(len(X_dig)):
            for x in X_dig[i]:
                for y in X_dig[j]:
                    X_left.append(x); X_right.append(y); Y.append([1] if i == j else [0])
    return [np.stack(X_left), np.stack(X_right)], np.stack(Y)


def roundoff(Y):
    """reference code/helpers.py:39-46"""
    return np.stack([[1] if y >= 0.5 else [0] for y in Y])


def select_queries(ens, disguised, batch_y, col, disparity_ratio, eps, blind_strategy=False):
    """reference code/ALINK_arc.py:167-198 (col 0) / code/ALINK.py:170-201 (col 1), python loops kept.
    Returns (set of query indices, ACTIVE_COUNT increment)."""
    mis = []
    for dp in disguised:
        disparities = []
        for j in range(len(dp)):
            c1 = dp[j][col]
            c2 = ens[j][col]
            if blind_strategy:
                if (c1 >= 0.5) != (c2 >= 0.5):
                    disparities.append(j)
            else:
                disparities.append(-np.absolute(c1 - c2))
        if not blind_strategy:
            disparities = np.argsort(disparities, kind="stable")[:int(len(disparities) * disparity_ratio)]
        mis.append(disparities)
    works = set(int(i) for i in mis[0])
    for j in range(1, len(mis)):
        works = works & set(int(i) for i in mis[j])
    query, active = [], 0
    for j in works:
        e = ens[j][col]
        if e <= 0.5 - eps or e >= 0.5 + eps:
            c1 = e >= 0.5
            c2 = batch_y[j][0] >= 0.5
            active += 1
            if c1 == c2:
                query.append(j)
    return set(query), active

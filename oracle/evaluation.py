"""oracle/evaluation.py — NumPy restatement of the DFW evaluation utilities.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/eval_roc.npz and
eval_stats.npz, produced by RUNNING the reference's own scripts on seeded inputs
(tests/golden/make_golden_eval.py):
    score matrix   reference utilities/generateMatrixDFW.py:25-36   (scores[i][j] = predict([f_i, f_j])[0])
    TPR / FPR      reference utilities/ROC_precompute.py:19-66
    AUC/EER/GAR    reference utilities/getStats.py:5-25
"""
import numpy as np


def score_matrix(predict, features, col=0):
    """generateMatrixDFW.py:28-35: row i = predict([tile(f_i), features])[:, col]."""
    rows = []
    for i in range(len(features)):
        left = np.repeat(features[i][None], len(features), axis=0)
        rows.append(np.asarray(predict([left, features]))[:, col])
    return np.stack(rows)


def genuine_impostor(score_matrix, mask, roc_case):
    """ROC_precompute.py:24-44: strict upper triangle, row-major order."""
    n = score_matrix.shape[0]
    iu = np.triu_indices(n, k=1)
    s, m = np.asarray(score_matrix)[iu], np.asarray(mask)[iu]
    if roc_case == 3:
        g, i = (m == 1) | (m == 2), (m == 3) | (m == 4)
    elif roc_case == 2:
        g, i = m == 2, m == 4
    elif roc_case == 1:
        g, i = m == 1, m == 3
    else:
        raise ValueError("roc_case must be 1, 2 or 3")
    return s[g].astype(np.float64), s[i].astype(np.float64)


def roc_precompute(score_matrix, mask, thresholds, roc_case):
    """ROC_precompute.py:51-66 -> (true_positive_rate, false_positive_rate), one entry per threshold
    in the order given.  Python-3 true division (`from __future__ import division`, :1)."""
    gen, imp = genuine_impostor(score_matrix, mask, roc_case)
    thr = np.asarray(thresholds, dtype=np.float64)
    tp = np.array([(gen >= t).sum() for t in thr])
    fp = np.array([(imp >= t).sum() for t in thr])
    return tp / len(gen), fp / len(imp)


def find_nearest(array, value):
    return (np.abs(array - value)).argmin()


def auc(x, y):
    """sklearn.metrics.auc (third-party; trapezoidal rule, x monotonic in either direction)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    dx = np.diff(x)
    direction = 1
    if np.any(dx < 0):
        if np.all(dx <= 0):
            direction = -1
        else:
            raise ValueError("x is neither increasing nor decreasing : {}.".format(x))
    return direction * np.sum(dx * (y[1:] + y[:-1]) / 2.0)


def get_stats(TPR, FPR):
    """getStats.py:9-25 -> [AUC, EER, GAR@1%FAR, GAR@0.1%FAR]."""
    TPR, FPR = np.asarray(TPR, np.float64), np.asarray(FPR, np.float64)
    FNR = 1 - TPR
    eer = FPR[np.nanargmin(np.absolute(FNR - FPR))]
    return np.array([auc(FPR, TPR), eer, TPR[find_nearest(FPR, 0.010)], TPR[find_nearest(FPR, 0.0010)]])

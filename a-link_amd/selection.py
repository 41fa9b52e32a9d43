"""selection — A-LINK's query-selection rule, which the reference keeps inline in its drivers.

select_queries restates reference code/ALINK_arc.py:167-198 (column 0) and code/ALINK.py:170-201
(column 1): per noise, disparity d_j = -|M2_noise[j][c] - M1[j][c]| and the int(P*disparity_ratio)
smallest are kept (or, blind strategy, those whose decisions differ); intersect over noises; drop
the grey band |M1 - 0.5| < eps; count oracle queries; keep those where the ensemble agrees with the
oracle.  Host NumPy like the reference (P is a few thousand per iteration); for pool-scale P the
top-k runs on device through alink_score / alink_topk (see pool.py).

Tie order: np.argsort (quicksort) is not a stable contract, and Python-2 `Set` iteration order is
arbitrary, so the reference's *ordering* of queryIndices is not reproducible in principle; the SET is.
Here ties break towards the lower index (stable sort) and queryIndices come out ascending.
"""
import numpy as np

from .helpers import roundoff


def disparity_indices(dp, ens, col, disparity_ratio, blind_strategy=False):
    dp = np.asarray(dp)
    ens = np.asarray(ens)
    if blind_strategy:
        c1 = dp[:, col] >= 0.5
        c2 = ens[:, col] >= 0.5
        return np.nonzero(c1 != c2)[0]
    d = -np.absolute(dp[:, col] - ens[:, col])
    k = int(len(d) * disparity_ratio)
    return np.argsort(d, kind="stable")[:k]


def select_queries(ensemblePredictions, disguisedPredictions, batch_y, col=0, disparity_ratio=0.25, eps=0.05,
                   blind_strategy=False):
    """Returns (queryIndices ascending, active_count, labels) — labels = roundoff(M1[q][col])."""
    ens = np.asarray(ensemblePredictions)
    sets = [set(disparity_indices(dp, ens, col, disparity_ratio, blind_strategy).tolist())
            for dp in disguisedPredictions]
    works = sets[0]
    for s in sets[1:]:
        works = works & s
    queryIndices, active = [], 0
    for j in sorted(works):
        e = ens[j][col]
        if e <= 0.5 - eps or e >= 0.5 + eps:
            c1 = e >= 0.5
            c2 = batch_y[j][0] >= 0.5
            active += 1
            if c1 == c2:
                queryIndices.append(j)
    labels = roundoff(ens[queryIndices, col]) if queryIndices else np.zeros((0, 1), dtype=int)
    return queryIndices, active, labels


def partition_by_noise(queryIndices, n_noise):
    """reference code/ALINK_arc.py:213-222: chunk k of size mp = int(len(q)/n_noise) takes noise k."""
    mp = int(len(queryIndices) / float(n_noise))
    return [queryIndices[i * mp:(i + 1) * mp] for i in range(n_noise)]

"""Shared builders for the selection-parity tests (tests/test_gpu_pool.py, tools/selection_flips.py): separable
synthetic identities, oracle-side head training, and the set of pairs whose selection may legitimately differ
between two runs whose scores differ by at most a MEASURED delta."""
import numpy as np

from oracle import calibrate
from oracle import siamese_head as O


def identities(n_persons, counts, size, seed, var=45.0):
    """Per person one smooth base 'face' (oracle.calibrate.calibration_pixels) and counts[p] images of it:
    base + N(0, var) pixel noise + N(0, 12) brightness, clipped to 0..255 float32 (what readDFW hands the
    models: float32 RGB 0..255, reference code/readDFW.py:82).  Returns a list of (k, H, W, 3) arrays."""
    rng = np.random.default_rng(seed)
    bases = calibrate.calibration_pixels(n_persons, size, seed=seed)
    out = []
    for p in range(n_persons):
        k = counts[p]
        imgs = bases[p][None] + rng.normal(0, var, (k,) + tuple(size) + (3,)) + rng.normal(0, 12, (k, 1, 1, 1))
        out.append(np.clip(imgs, 0, 255).astype(np.float32))
    return out


def unique_rows(people, n_plain):
    """createMiniBatchIndices' row order: every person's plain images, then every person's disguised ones."""
    return np.concatenate([p[:a] for p, a in zip(people, n_plain)] + [p[a:] for p, a in zip(people, n_plain)])


def train_head(seed, E, li, ri, y, epochs, lr=1.0, batch_size=16, d_in=512):
    """An oracle HeadModel fine-tuned the way the reference does it (SiameseNetwork.finetune, reference
    code/siamese.py:52-58: fit with validation_split 0.2) on pairs gathered from embedding matrix E."""
    m = O.HeadModel(d_in, lr=lr, seed=seed)
    np.random.seed(seed)
    O.finetune(m, [E[li], E[ri]], y, epochs, batch_size)
    return m


def balanced_subset(y, neg_per_pos, seed):
    rng = np.random.default_rng(seed)
    pos = np.flatnonzero(y[:, 0] == 1)
    neg = rng.choice(np.flatnonzero(y[:, 0] == 0), len(pos) * neg_per_pos, replace=False)
    return rng.permutation(np.concatenate([pos, neg]))


def _kth_smallest(v, k):
    return np.sort(v, kind="stable")[k - 1]


def selection_fragile(ens_o, dis_o, ens_g, dis_g, col, disparity_ratio, eps):
    """Pairs whose membership in the A-LINK query set (reference code/ALINK_arc.py:167-198) may legitimately differ
    between the oracle run (ens_o, dis_o) and the device run (ens_g, dis_g), from the MEASURED differences.

    Rule, per noise n: disparity d = -|dis_n - ens| (column col), keep the k = int(P * ratio) smallest; intersect over
    noises; drop the grey band |ens - 0.5| < eps; compare the decision ens >= 0.5 with the label.
    If pair j is among the k smallest for one run and not the other then (with t the k-th smallest value of each run)
        |d_o[j] - t_o| <= |d_g[j] - d_o[j]| + |t_g - t_o|
    — its own measured error plus the measured shift of the cut; likewise the grey-band and decision tests change only
    where |ens_o - (0.5 -+ eps)| resp. |ens_o - 0.5| is within the pair's own |ens_g - ens_o|.  Returns
    (per_pair, uniform): the set defined by each pair's own error, and the superset obtained with the MAXIMUM error of
    any pair in place of the pair's own (what a caller who only knows "probabilities agree to delta" can promise)."""
    P = len(ens_o)
    k = int(P * disparity_ratio)
    eo, eg = ens_o[:, col].astype(np.float64), ens_g[:, col].astype(np.float64)
    e_ens = np.abs(eg - eo)
    per_pair, uniform = np.zeros(P, bool), np.zeros(P, bool)
    for do, dg in zip(dis_o, dis_g):
        d_o = -np.abs(do[:, col].astype(np.float64) - eo)
        d_g = -np.abs(dg[:, col].astype(np.float64) - eg)
        if not 0 < k <= P:
            continue
        t_o, t_g = _kth_smallest(d_o, k), _kth_smallest(d_g, k)
        err = np.abs(d_g - d_o)
        margin = np.abs(d_o - t_o)
        per_pair |= margin <= err + abs(t_g - t_o)
        uniform |= margin <= 2 * err.max()
    for edge in (0.5 - eps, 0.5, 0.5 + eps):
        per_pair |= np.abs(eo - edge) <= e_ens
        uniform |= np.abs(eo - edge) <= e_ens.max()
    return set(np.flatnonzero(per_pair).tolist()), set(np.flatnonzero(uniform).tolist())


def topk_fragile(scores, k, delta):
    """Indices whose membership in the top-k of `scores` can differ when every score moves by at most delta."""
    order = np.lexsort((np.arange(len(scores)), -scores))
    thr = scores[order[k - 1]]
    return set(np.flatnonzero(np.abs(scores - thr) <= 2 * delta).tolist())


def binary_entropy_topk_fragile(p0_oracle, p0_device, k, eps_entropy):
    """For two classes the entropy is a strictly decreasing function of u = |p0 - 1/2|, so "the k most uncertain" is
    "the k smallest u".  With c the k-th smallest u of each run, pair j can be in one run's set and not the other's only
    if |u_o[j] - c_o| <= |u_g[j] - u_o[j]| + |c_g - c_o| (+ the selecting side's own entropy-evaluation error, which
    near the cut is worth eps_entropy / c in u since dH/du ~ -4u; doubled for safety).  Returns
    (per_pair set, uniform set [max error instead of the pair's own], c_o)."""
    uo = np.abs(np.asarray(p0_oracle, np.float64) - 0.5)
    ug = np.abs(np.asarray(p0_device, np.float64) - 0.5)
    c_o, c_g = _kth_smallest(uo, k), _kth_smallest(ug, k)
    eta = 2 * eps_entropy / max(c_o, 1e-6)
    err = np.abs(ug - uo)
    margin = np.abs(uo - c_o)
    per_pair = margin <= err + abs(c_g - c_o) + eta
    uniform = margin <= 2 * err.max() + eta
    return set(np.flatnonzero(per_pair).tolist()), set(np.flatnonzero(uniform).tolist()), float(c_o)

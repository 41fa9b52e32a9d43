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


def selection_fragile(ens, dis, col, disparity_ratio, eps, delta_ens, delta_dis):
    """Pairs whose membership in the A-LINK query set (reference code/ALINK_arc.py:167-198) can differ between the
    oracle run (ens, dis) and a run whose ensemble / disguised probabilities differ from them by at most
    delta_ens / delta_dis: a disparity d = -|dis - ens| moves by at most delta_ens + delta_dis, and so does the
    k-th smallest one, so only |d_j - d_(k)| <= 2 (delta_ens + delta_dis) can change sides; the grey-band and the
    decision test move only for |ens - (0.5 -+ eps)| <= delta_ens and |ens - 0.5| <= delta_ens."""
    P = len(ens)
    k = int(P * disparity_ratio)
    frag = set()
    dd = delta_ens + delta_dis
    for d in dis:
        disp = -np.abs(d[:, col] - ens[:, col])
        if 0 < k <= P:
            thr = np.sort(disp, kind="stable")[k - 1]
            frag |= set(np.flatnonzero(np.abs(disp - thr) <= 2 * dd).tolist())
    e = ens[:, col]
    frag |= set(np.flatnonzero(np.abs(np.abs(e - 0.5) - eps) <= delta_ens).tolist())
    frag |= set(np.flatnonzero(np.abs(e - 0.5) <= delta_ens).tolist())
    return frag


def topk_fragile(scores, k, delta):
    """Indices whose membership in the top-k of `scores` can differ when every score moves by at most delta."""
    order = np.lexsort((np.arange(len(scores)), -scores))
    thr = scores[order[k - 1]]
    return set(np.flatnonzero(np.abs(scores - thr) <= 2 * delta).tolist())

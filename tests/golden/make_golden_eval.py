#!/usr/bin/env python3
"""Generates tests/golden/eval_*.npz by RUNNING the reference's own evaluation scripts
(/root/reference/utilities/ROC_precompute.py and getStats.py — this container only) on seeded
synthetic inputs and recording what they write / print.  Only data is stored.

ROC_precompute.py hard-codes the DFW test-set size (7771 x 7771, utilities/ROC_precompute.py:24-26),
so its inputs cannot be committed (2 x 60 M numbers): they are a pure function of the seed below and
tests regenerate them with `eval_inputs(seed)` from this file; the fixture holds the seed, the
thresholds and the TPR/FPR the reference wrote.

Run:  python tests/golden/make_golden_eval.py        (about 5 minutes, 3 GB of /tmp)
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF_UTIL = "/root/reference/utilities"
HERE = os.path.dirname(os.path.abspath(__file__))
N_DFW = 7771


def eval_inputs(seed, n=N_DFW):
    """Seeded score matrix (multiples of 1/1024: exact in float32, float64 and 10-decimal text, so the
    reference's float64 comparisons and the device's widened-float32 ones see the same numbers — and
    0, 0.25, 0.5, 0.75, 1 tie with thresholds exactly), protocol mask
    (0 = unused pair, 1/2 genuine, 3/4 impostor — utilities/ROC_precompute.py:27-44) and thresholds."""
    rng = np.random.default_rng(seed)
    scores = rng.integers(0, 1025, (n, n)).astype(np.float64) / 1024.0
    mask = rng.choice(np.arange(5, dtype=np.int64), size=(n, n), p=[0.98, 0.004, 0.004, 0.006, 0.006])
    thr = np.concatenate([np.linspace(0.0, 1.0, 21), [0.3333, 0.5005, 1.5, -0.25]])   # unsorted on purpose
    return scores, mask, thr


def _write_matrix(path, a, fmt):
    with open(path, "w") as f:
        for row in a:
            f.write(" ".join(fmt % v for v in row))
            f.write("\n")


def run_roc_precompute(seed):
    scores, mask, thr = eval_inputs(seed)
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        _write_matrix(os.path.join(d, "scores.txt"), scores, "%.10f")
        _write_matrix(os.path.join(d, "updated_testing_mask.txt"), mask, "%d")
        np.savetxt(os.path.join(d, "thresholds.txt"), thr)
        for case in (1, 2, 3):
            subprocess.check_call([sys.executable, os.path.join(REF_UTIL, "ROC_precompute.py"), "scores.txt",
                                   "roc%d.txt" % case, str(case)], cwd=d, env=dict(os.environ, MPLBACKEND="Agg"))
            out["case%d" % case] = np.loadtxt(os.path.join(d, "roc%d.txt" % case))
    return thr, out


def run_get_stats(tpr, fpr):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        np.savetxt(os.path.join(d, "roc.txt"), np.array([tpr, fpr]))
        txt = subprocess.check_output([sys.executable, os.path.join(REF_UTIL, "getStats.py"), "roc.txt"], cwd=d).decode()
    auc = float(re.search(r"AUC ([-\d.einf]+)", txt).group(1))
    eer = float(re.search(r"EER ([-\d.einf]+)", txt).group(1))
    gars = [float(x) for x in re.findall(r"GAR is ([-\d.einf]+) for", txt)]
    return np.array([auc, eer, gars[0], gars[1]])


def main():
    seed = 20261003
    thr, roc = run_roc_precompute(seed)
    np.savez(os.path.join(HERE, "eval_roc.npz"), seed=seed, thresholds=thr, **roc)
    # getStats on (i) the three ROC curves above (thresholds sorted so FPR is monotonic, as DFW's
    # thresholds.txt is) and (ii) a smooth synthetic curve
    stats_in, stats_out = [], []
    order = np.argsort(thr)
    for case in (1, 2, 3):
        tpr, fpr = roc["case%d" % case][0][order], roc["case%d" % case][1][order]
        stats_in.append(np.stack([tpr, fpr]))
        stats_out.append(run_get_stats(tpr, fpr))
    t = np.linspace(0, 1, 2001)
    fpr = (1 - t) ** 3
    tpr = 1 - t ** 2.5
    stats_in.append(np.stack([tpr, fpr]))
    stats_out.append(run_get_stats(tpr, fpr))
    np.savez(os.path.join(HERE, "eval_stats.npz"), **{"curve%d" % i: a for i, a in enumerate(stats_in)},
             stats=np.stack(stats_out))       # rows: [auc, eer, gar@1%, gar@0.1%] as printed (%f: 6 decimals)
    print("wrote eval_roc.npz, eval_stats.npz")


if __name__ == "__main__":
    main()

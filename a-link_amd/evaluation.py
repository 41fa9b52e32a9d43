"""evaluation — the DFW evaluation utilities of the reference on the GPU.

    score_matrix(model, features)                 utilities/generateMatrixDFW.py:25-36
    roc_precompute(scores, mask, thresholds, c)   utilities/ROC_precompute.py:19-66
    get_stats(TPR, FPR)                           utilities/getStats.py:5-25

The N x N matrix is the same pair scorer (head.hip) with the pairs enumerated by the kernel
(alink_pair_scores_matrix); the genuine/impostor threshold counts are one streaming pass over the
matrix and the protocol mask (alink_roc_counts, evaluate.hip).  The T-sized tail (suffix sums, rates,
AUC/EER/GAR) is host NumPy like the reference's.
"""
import ctypes as C

import numpy as np

from . import _abi


def _heads_of(model):
    """DenseHead handles behind a SiameseNetwork, a Bagging of them, or a bare DenseHead / list."""
    from .head import DenseHead
    if isinstance(model, DenseHead):
        return [model]
    if isinstance(model, (list, tuple)):
        return [h for m in model for h in _heads_of(m)]
    if hasattr(model, "models"):
        return [h for m in model.models for h in _heads_of(m)]
    net = getattr(model, "siamese_net", None)
    if isinstance(net, DenseHead):
        return [net]
    raise TypeError("score_matrix needs DenseHead-backed models, got %r" % (model,))


def score_matrix(model, features, col=0, out=None, rows_per_call=1024):
    """scores[i][j] = model.predict([features[i], features[j]])[col]  (generateMatrixDFW.py:28-35; the
    reference keeps column 0).  `model`: SiameseNetwork / DenseHead, or a Bagging / list of them
    (member mean, code/committee.py:13-20).  NumPy in -> NumPy out, CUDA tensor in -> CUDA tensor out."""
    heads = _heads_of(model)
    h0 = heads[0]
    torch = h0.torch
    as_torch = isinstance(features, torch.Tensor)
    E = h0._dev(features)
    n = E.shape[0]
    assert E.ndim == 2 and E.shape[1] == h0.d_in, "features must be (N, %d)" % h0.d_in
    if out is None:
        out = torch.empty((n, n), dtype=torch.float32, device=h0.device)
    assert out.shape == (n, n) and out.dtype == torch.float32 and out.is_contiguous()
    arr = (C.c_void_p * len(heads))(*[h.h for h in heads])
    rows_per_call = max(1, min(int(rows_per_call), (1 << 31) // max(n, 1)))
    for r0 in range(0, n, rows_per_call):
        nr = min(rows_per_call, n - r0)
        _abi.check(h0.lib.alink_pair_scores_matrix(arr, len(heads), _abi.ptr(E), n, r0, nr, int(col),
                                                   C.c_void_p(out.data_ptr() + 4 * r0 * n),
                                                   _abi.current_stream()), "alink_pair_scores_matrix")
    return out if as_torch else out.cpu().numpy()


def roc_counts(score_matrix, mask, thresholds, roc_case, device=None):
    """(true_positive, false_positive, n_genuine, n_impostor): integer counts per threshold (in the
    order given) over the strict upper triangle — the loop bodies of ROC_precompute.py:24-61."""
    import torch
    device = _abi.resolve_device(device)
    lib = _abi.init(device)
    dev = "cuda:%d" % device
    S = score_matrix if isinstance(score_matrix, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(score_matrix, dtype=np.float32))
    S = S.to(dev, torch.float32).contiguous()
    M = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask).astype(np.uint8))
    M = M.to(dev, torch.uint8).contiguous()
    n = S.shape[0]
    assert S.shape == (n, n) and M.shape == (n, n), "score matrix and mask must both be N x N"
    if roc_case not in (1, 2, 3):
        raise ValueError("Son, you screwed up.")          # ROC_precompute.py:45-47
    thr = np.asarray(thresholds, dtype=np.float64).ravel()
    T = thr.size
    order = np.argsort(thr, kind="stable")
    thr_d = torch.from_numpy(np.ascontiguousarray(thr[order])).to(dev)
    hist = torch.empty((2, T + 1), dtype=torch.int64, device=dev)
    _abi.check(lib.alink_roc_counts(_abi.ptr(S), _abi.ptr(M), n, _abi.ptr(thr_d), T, int(roc_case), _abi.ptr(hist),
                                    _abi.current_stream()), "alink_roc_counts")
    h = hist.cpu().numpy()
    # scores with exactly c sorted thresholds <= score: >= threshold of rank t  <=>  c > t
    suffix = np.cumsum(h[:, ::-1], axis=1)[:, ::-1]          # suffix[k][c] = sum_{c' >= c} h[k][c']
    tp_sorted, fp_sorted = suffix[0, 1:], suffix[1, 1:]
    tp, fp = np.empty(T, np.int64), np.empty(T, np.int64)
    tp[order], fp[order] = tp_sorted, fp_sorted
    return tp, fp, int(h[0].sum()), int(h[1].sum())


def roc_precompute(score_matrix, mask, thresholds, roc_case, device=None):
    """-> (true_positive_rate, false_positive_rate) as ROC_precompute.py:51-66 saves them."""
    tp, fp, ng, ni = roc_counts(score_matrix, mask, thresholds, roc_case, device)
    if ni == 0 or ng == 0:
        raise ZeroDivisionError("division by zero")       # what ROC_precompute.py:60-62 does with an empty class
    return tp / np.float64(ng), fp / np.float64(ni)


def find_nearest(array, value):
    idx = (np.abs(array - value)).argmin()
    return idx


def auc(x, y):
    """sklearn.metrics.auc as called at getStats.py:13: trapezoidal area, x monotonic either way."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if x.shape[0] < 2:
        raise ValueError("At least 2 points are needed to compute area under curve, but x.shape = %s" % (x.shape,))
    dx = np.diff(x)
    direction = 1
    if np.any(dx < 0):
        if np.all(dx <= 0):
            direction = -1
        else:
            raise ValueError("x is neither increasing nor decreasing : {}.".format(x))
    return direction * np.sum(dx * (y[1:] + y[:-1]) / 2.0)


def get_stats(TPR, FPR, verbose=False):
    """getStats.py:9-25 -> dict(auc, eer, gar_at_1pct_far, gar_at_0p1pct_far)."""
    TPR, FPR = np.asarray(TPR, np.float64), np.asarray(FPR, np.float64)
    FNR = 1 - TPR
    eer = FPR[np.nanargmin(np.absolute(FNR - FPR))]
    out = {"auc": auc(FPR, TPR), "eer": eer,
           "gar_at_1pct_far": TPR[find_nearest(FPR, 0.010)],
           "gar_at_0p1pct_far": TPR[find_nearest(FPR, 0.0010)]}
    if verbose:
        print("AUC %f" % out["auc"])
        print("EER %f" % out["eer"])
        print('GAR is %f for %f FAR' % (out["gar_at_1pct_far"], 0.010))
        print('GAR is %f for %f FAR' % (out["gar_at_0p1pct_far"], 0.0010))
    return out


def main(argv=None):
    """The three evaluation scripts behind one entry point (same positional arguments as the originals):
        python -m a_link_amd.evaluation matrix MODELNAME OUTPUT      utilities/generateMatrixDFW.py (processedData.npy in cwd)
        python -m a_link_amd.evaluation roc SCORES OUTPUT ROC_CASE   utilities/ROC_precompute.py (mask + thresholds in cwd)
        python -m a_link_amd.evaluation stats ROCFILE                utilities/getStats.py
    """
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("matrix", "roc", "stats"):
        print(main.__doc__)
        return 2
    cmd, args = argv[0], argv[1:]
    if cmd == "stats":
        TPR, FPR = np.loadtxt(args[0])
        get_stats(TPR, FPR, verbose=True)
        return 0
    if cmd == "roc":
        scores = np.loadtxt(args[0], dtype=float)
        mask = np.loadtxt('updated_testing_mask.txt', dtype=int)
        thresholds = np.loadtxt('thresholds.txt', dtype=float)
        tpr, fpr = roc_precompute(scores, mask, thresholds, int(args[2]))
        print('Genuine and Imposter score generated')
        np.savetxt(args[1], np.array([tpr, fpr]))
        return 0
    from . import siamese
    features = np.load("processedData.npy")
    model = siamese.SiameseNetwork((features.shape[1],), args[0], 0.1)
    if model.maybeLoadFromMemory():
        print("Loaded model successfully!")
    else:
        print("Oops! Model not found")
        return 1
    np.savetxt(args[1], score_matrix(model, features, col=0))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

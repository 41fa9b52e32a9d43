"""uncertainty — drop-in for reference code/uncertainty.py (vendored modAL, patched for pair input).

Measures (code/uncertainty.py:15-60) and samplers (code/uncertainty.py:133-217) with the same
signatures.  Small inputs use the same NumPy expressions as the reference (host logic, like the
reference); `*_device` variants score and top-k a pool resident on the GPU through
alink_score / alink_topk (select.hip) — ties break towards the lower index there.

Kept quirk: the samplers return `[X[0][idx], X[0][idx]]` — the LEFT array twice
(code/uncertainty.py:159,187,217).
"""
import numpy as np
from scipy.stats import entropy

try:  # sklearn is present in this image; keep the reference's exception type when it is
    from sklearn.exceptions import NotFittedError
except Exception:  # pragma: no cover
    class NotFittedError(Exception):
        pass


def multi_argmax(values, n_instances=1):
    """modAL.utils.selection.multi_argmax: indices of the n largest values."""
    assert n_instances <= values.shape[0], 'n_instances must be less or equal than the size of utility'
    return np.argpartition(-values, n_instances - 1, axis=0)[:n_instances]


def shuffled_argmax(values, n_instances=1):
    """modAL.utils.selection.shuffled_argmax: random tie-break by shuffling first."""
    assert n_instances <= values.shape[0], 'n_instances must be less or equal than the size of utility'
    shuffled_idx = np.random.permutation(len(values))
    shuffled_values = values[shuffled_idx]
    sorted_query_idx = np.argsort(shuffled_values, kind='mergesort')[len(shuffled_values) - n_instances:]
    return shuffled_idx[sorted_query_idx]


def _proba_uncertainty(proba):
    return 1 - np.max(proba, axis=1)


def _proba_margin(proba):
    if proba.shape[1] == 1:
        return np.zeros(shape=len(proba))
    part = np.partition(-proba, 1, axis=1)
    margin = - part[:, 0] + part[:, 1]
    return margin


def _proba_entropy(proba):
    return np.transpose(entropy(np.transpose(proba)))


def classifier_uncertainty(classifier, X, **predict_proba_kwargs):
    try:
        classwise_uncertainty = classifier.predict_proba(X, **predict_proba_kwargs)
    except NotFittedError:
        return np.ones(shape=(X[0].shape[0], ))
    return 1 - np.max(classwise_uncertainty, axis=1)


def classifier_margin(classifier, X, **predict_proba_kwargs):
    try:
        classwise_uncertainty = classifier.predict_proba(X, **predict_proba_kwargs)
    except NotFittedError:
        return np.zeros(shape=(X[0].shape[0], ))
    if classwise_uncertainty.shape[1] == 1:
        return np.zeros(shape=(classwise_uncertainty.shape[0],))
    part = np.partition(-classwise_uncertainty, 1, axis=1)
    return -part[:, 0] + part[:, 1]


def classifier_entropy(classifier, X, **predict_proba_kwargs):
    try:
        classwise_uncertainty = classifier.predict_proba(X, **predict_proba_kwargs)
    except NotFittedError:
        return np.zeros(shape=(X[0].shape[0], ))
    return np.transpose(entropy(np.transpose(classwise_uncertainty)))


def _pick(utility, n_instances, random_tie_break):
    if not random_tie_break:
        return multi_argmax(utility, n_instances=n_instances)
    return shuffled_argmax(utility, n_instances=n_instances)


def uncertainty_sampling(classifier, X, n_instances=1, random_tie_break=False, **uncertainty_measure_kwargs):
    uncertainty = classifier_uncertainty(classifier, X, **uncertainty_measure_kwargs)
    query_idx = _pick(uncertainty, n_instances, random_tie_break)
    return query_idx, [X[0][query_idx], X[0][query_idx]]


def margin_sampling(classifier, X, n_instances=1, random_tie_break=False, **uncertainty_measure_kwargs):
    margin = classifier_margin(classifier, X, **uncertainty_measure_kwargs)
    query_idx = _pick(-margin, n_instances, random_tie_break)
    return query_idx, [X[0][query_idx], X[0][query_idx]]


def entropy_sampling(classifier, X, n_instances=1, random_tie_break=False, **uncertainty_measure_kwargs):
    ent = classifier_entropy(classifier, X, **uncertainty_measure_kwargs)
    query_idx = _pick(ent, n_instances, random_tie_break)
    return query_idx, [X[0][query_idx], X[0][query_idx]]


# ---- device variants: pool-scale scoring + top-k on the GPU ------------------------------------------
_KIND = {"uncertainty": 0, "margin": 1, "entropy": 2}


def score_device(probs, kind, b=None, col=0):
    """probs: CUDA (P,C) f32 tensor -> CUDA (P,) scores (alink_score)."""
    import torch
    from . import _abi
    lib = _abi.init(probs.device.index or 0)
    code = _KIND[kind] if kind in _KIND else _abi.SCORE_DISPARITY
    probs = probs.contiguous()
    P, Cn = probs.shape
    out = torch.empty(P, dtype=torch.float32, device=probs.device)
    _abi.check(lib.alink_score(code, _abi.ptr(probs), _abi.ptr(b.contiguous() if b is not None else None), col, P, Cn,
                               _abi.ptr(out), _abi.current_stream(probs.device)), "alink_score")
    return out


def topk_device(scores, k, largest=True):
    """indices (int32 CUDA) of the k largest/smallest scores, sorted, ties -> lower index."""
    import ctypes as C
    import torch
    from . import _abi
    lib = _abi.init(scores.device.index or 0)
    scores = scores.contiguous()
    P = scores.numel()
    nb = lib.alink_topk_scratch_bytes(P, int(k))
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device=scores.device)
    off = (-scratch.data_ptr()) % 256
    idx = torch.empty(int(k), dtype=torch.int32, device=scores.device)
    vals = torch.empty(int(k), dtype=torch.float32, device=scores.device)
    _abi.check(lib.alink_topk(_abi.ptr(scores), P, int(k), 1 if largest else 0, _abi.ptr(idx), _abi.ptr(vals),
                              C.c_void_p(scratch.data_ptr() + off), _abi.current_stream(scores.device)), "alink_topk")
    return idx, vals


# ---- screen-then-settle forms of the three samplers (round 4; a-link_amd/settle.py) -----------------------------------------
def _sampling_settled(kind, largest, classifier, X_screen, exact_rows, n_instances, **settle_kw):
    """The sampler's query_idx (as a set: the order inside the top-n is not contractual, reference code/uncertainty.py uses modAL's
    argpartition) from SCREENED pair features, identical to the sampler run on exact features.

    classifier   the pair scorer (`predict_proba([L, R])` -> (P, 2)); it is evaluated exactly — what is screened is its INPUT
    X_screen     [L, R] pair features from the 16-bit screening mode of the feature model (`ArcFace.process_screen`)
    exact_rows   f(sorted pair indices) -> [L, R] exact features of those pairs (re-embeds their images in the exact mode)
    Pairs whose side of the n-th cut is uncertain under the measured screening error are settled, the rest keep their side."""
    from . import settle as _settle
    proba_s = np.asarray(classifier.predict_proba(X_screen))
    if proba_s.shape[1] != 2:
        raise ValueError("screen-then-settle sampling is built for two-class pair scorers (got %d classes)" % proba_s.shape[1])
    score = {"uncertainty": _proba_uncertainty, "margin": _proba_margin, "entropy": _proba_entropy}[kind]
    P = len(proba_s)

    def exact_fn(rows):
        pr = np.asarray(classifier.predict_proba(exact_rows(rows)))
        return rows, pr[:, 0], score(pr).astype(np.float32)
    vals, idx, info = _settle.settle_topk(proba_s[:, 0], score(proba_s).astype(np.float32), np.arange(P), P, exact_fn, n_instances,
                                          kind=kind, largest=largest, **settle_kw)
    return np.asarray(idx), info


def uncertainty_sampling_settled(classifier, X_screen, exact_rows, n_instances=1, **settle_kw):
    return _sampling_settled("uncertainty", True, classifier, X_screen, exact_rows, n_instances, **settle_kw)


def margin_sampling_settled(classifier, X_screen, exact_rows, n_instances=1, **settle_kw):
    return _sampling_settled("margin", False, classifier, X_screen, exact_rows, n_instances, **settle_kw)


def entropy_sampling_settled(classifier, X_screen, exact_rows, n_instances=1, **settle_kw):
    return _sampling_settled("entropy", True, classifier, X_screen, exact_rows, n_instances, **settle_kw)

"""CPU: two pieces of the oracle restate a THIRD-PARTY call the reference makes, and that dependency IS installed here
(sklearn 1.7.2) — so they are checked against the dependency itself, on the reference's literal calls:

    oracle.ir_resnet.l2_normalize   sklearn.preprocessing.normalize(embedding)      reference code/face_model.py:92
    oracle.evaluation.auc           sklearn.metrics.auc(FPR, TPR)                   reference utilities/getStats.py:13

(the product's `l2` epilogue against the same call on the GPU: tests/test_gpu_backbone.py)."""
import numpy as np
import pytest

sklearn = pytest.importorskip("sklearn")
from sklearn import metrics, preprocessing      # noqa: E402

from oracle import evaluation as OE, ir_resnet as OI      # noqa: E402


def test_l2_normalize_is_sklearn_normalize():
    rs = np.random.RandomState(0)
    for dtype in (np.float32, np.float64):
        e = (rs.randn(9, 512) * 3).astype(dtype)
        e[3] = 0                                               # a zero row: sklearn leaves it zero (norm 0 -> 1)
        e[5] *= 1e-3                                           # a small one (norm ~0.07)
        # (a row whose norm is below 10 * eps of its dtype is a version matter: scikit-learn 0.2x — what the reference's era
        # installs — replaces only EXACT zeros by 1, as the oracle does; 1.x also leaves such rows undivided.  Not tested.)
        got = OI.l2_normalize(e)
        want = preprocessing.normalize(e)
        assert got.dtype == want.dtype == dtype
        np.testing.assert_allclose(got, want, rtol=3e-7 if dtype == np.float32 else 1e-15, atol=0)
        assert np.array_equal(got[3], np.zeros(512, dtype))
    # the reference's literal call shape: one embedding (1, 512) -> flatten
    one = rs.randn(1, 512).astype(np.float32)
    np.testing.assert_allclose(OI.l2_normalize(one).flatten(), preprocessing.normalize(one).flatten(), rtol=3e-7)


def test_auc_is_sklearn_auc_in_both_directions():
    rs = np.random.RandomState(1)
    for n in (2, 5, 300):
        fpr = np.sort(rs.rand(n))
        tpr = np.sort(rs.rand(n))
        assert OE.auc(fpr, tpr) == pytest.approx(metrics.auc(fpr, tpr), rel=1e-14, abs=1e-16)
        # thresholds ascending make FPR / TPR DEcrease (utilities/ROC_precompute.py:51-66): the other monotone direction
        assert OE.auc(fpr[::-1], tpr[::-1]) == pytest.approx(metrics.auc(fpr[::-1], tpr[::-1]), rel=1e-14, abs=1e-16)
    # repeated x values (ties in FPR) and the error for a non-monotone x
    x = np.array([0.0, 0.2, 0.2, 0.7, 1.0])
    y = np.array([0.0, 0.5, 0.6, 0.9, 1.0])
    assert OE.auc(x, y) == pytest.approx(metrics.auc(x, y), rel=1e-15)
    with pytest.raises(ValueError):
        OE.auc(np.array([0.0, 0.5, 0.3]), np.array([0.0, 0.5, 1.0]))
    with pytest.raises(ValueError):
        metrics.auc(np.array([0.0, 0.5, 0.3]), np.array([0.0, 0.5, 1.0]))


def test_get_stats_auc_entry_is_sklearn_auc_of_the_roc():
    """getStats.py:9-25 end to end on a synthetic ROC: the AUC entry equals sklearn's on the same (FPR, TPR)"""
    rs = np.random.RandomState(2)
    gen, imp = rs.rand(400) * 0.6 + 0.4, rs.rand(900) * 0.7
    thr = np.linspace(0, 1, 201)
    tpr = np.array([(gen >= t).mean() for t in thr])
    fpr = np.array([(imp >= t).mean() for t in thr])
    assert OE.get_stats(tpr, fpr)[0] == pytest.approx(metrics.auc(fpr, tpr), rel=1e-14)

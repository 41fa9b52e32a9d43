"""CPU: the modAL-style glue (base.py / learners.py) with a fake estimator — same call protocol as
reference code/existing_al.py:95-117 (query -> teach(only_new=True))."""
import numpy as np

import a_link_amd  # noqa: F401
from a_link_amd import learners, uncertainty


class FakeNet(object):
    """Keras-like model: predict -> (n,2) probabilities driven by the left features."""

    def __init__(self):
        self.fits = []

    def predict(self, X, **kw):
        s = 1.0 / (1.0 + np.exp(-X[0][:, 0]))
        return np.stack([1 - s, s], axis=1).astype(np.float32)

    def fit(self, X, y, **kw):
        self.fits.append((len(X[0]), y.shape, kw))


def test_active_learner_query_and_teach_only_new():
    net = FakeNet()
    wrapped = learners.KerasClassifier(lambda: net)
    learner = learners.ActiveLearner(estimator=wrapped, query_strategy=uncertainty.uncertainty_sampling)
    rng = np.random.RandomState(0)
    L = rng.randn(50, 4).astype(np.float32)
    R = rng.randn(50, 4).astype(np.float32)
    y = rng.randint(0, 2, (50, 1))
    # not fitted yet -> NotFittedError path -> uniform uncertainty of ones (code/uncertainty.py:77-80)
    idx, inst = learner.query([L, R], n_instances=5)
    assert len(idx) == 5
    learner.teach(X=[L[idx], R[idx]], y=y[idx], only_new=True, epochs=2, validation_split=0.1)
    assert net.fits[-1][0] == 5 and net.fits[-1][2] == {"epochs": 2, "validation_split": 0.1}
    idx2, inst2 = learner.query([L, R], n_instances=7)
    unc = 1 - net.predict([L, R]).max(axis=1)
    assert set(idx2.tolist()) == set(np.argsort(-unc, kind="stable")[:7].tolist())
    assert np.array_equal(inst2[0], L[idx2]) and np.array_equal(inst2[1], L[idx2])   # left twice (reference quirk)
    learner.teach(X=[L[idx2], R[idx2]], y=y[idx2])       # only_new=False -> refit on everything seen (5 + 7)
    assert net.fits[-1][0] == 12
    assert len(learner.X_training[0]) == 12 and len(learner.y_training) == 12


def test_committee_vote_proba_and_mean():
    nets = [FakeNet(), FakeNet()]
    ls = []
    for n in nets:
        w = learners.KerasClassifier(lambda n=n: n)
        w.model = n
        w.classes_ = np.array([0, 1])
        ls.append(learners.ActiveLearner(estimator=w))
    com = learners.Committee(ls, query_strategy=uncertainty.entropy_sampling)
    X = [np.linspace(-3, 3, 9)[:, None].astype(np.float32)] * 2
    vp = com.vote_proba(X)
    assert vp.shape == (9, 2, 2)
    np.testing.assert_allclose(com.predict_proba(X), nets[0].predict(X), atol=1e-7)
    assert np.array_equal(com.predict(X), (X[0][:, 0] > 0).astype(int))
    idx, _ = com.query(X, n_instances=3)
    assert set(idx.tolist()) == {3, 4, 5}


def test_default_precision_of_the_reference_api_classes():
    """The reference embeds in float32 (code/face_model.py:90): models built through its API without a dtype get the split
    precision mode, whose selection sets equal the f32 arithmetic's; the gradient pass exists for 16-bit storage only."""
    from a_link_amd import _abi, face_model
    assert face_model.default_dtype() == "f16x2"
    assert face_model.default_dtype(small_batch_split=True) == "f16x2"
    assert face_model.default_dtype(enable_grad=True) == "bf16"
    assert (_abi.DT_BF16, _abi.DT_F16, _abi.DT_F32, _abi.DT_F16X2) == (0, 1, 2, 3)
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "alink_hip.h")).read()
    for name, val in (("ALINK_DT_BF16", 0), ("ALINK_DT_F16", 1), ("ALINK_DT_F32", 2), ("ALINK_DT_F16X2", 3)):
        assert re.search(r"#define\s+%s\s+%d\b" % (name, val), hdr), name

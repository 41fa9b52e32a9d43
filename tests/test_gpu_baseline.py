"""GPU: the baseline scripts' pair scorer (siamese3: Dense(1, sigmoid), alink_head_create_ex out_dim=1)
against the NumPy oracle, and the modAL-style baseline loop of code/existing_al.py:95-121 through the
KerasClassifier facade."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(n, d, seed):
    rng = np.random.RandomState(seed)
    return rng.randn(n, d).astype(np.float32), rng.randn(n, d).astype(np.float32)


def test_sigmoid_head_matches_oracle(gpu):
    from a_link_amd import siamese3
    from oracle import siamese_head as O
    net = siamese3.SiameseNetwork((512,), "m", 0.1, seed=3)
    ws = net.siamese_net.get_weights()
    assert [w.shape for w in ws][-2:] == [(64, 1), (1,)]
    o = O.HeadModel(512, lr=0.1, out_dim=1)
    o.set_weights(ws)
    L, R = _data(300, 512, 0)
    p = net.predict([L, R])
    assert p.shape == (300, 1)
    np.testing.assert_allclose(p, o.predict([L, R]), atol=3e-6)
    y = (np.random.RandomState(1).rand(300, 1) > 0.5).astype(np.float32)
    for s in range(0, 96, 16):
        mg = net.siamese_net.train_on_batch([L[s:s + 16], R[s:s + 16]], y[s:s + 16])
        mo = o.train_on_batch([L[s:s + 16], R[s:s + 16]], y[s:s + 16])
        np.testing.assert_allclose(mg, mo, rtol=2e-5, atol=2e-6)
    for a, b in zip(net.siamese_net.get_weights(), o.get_weights()):
        np.testing.assert_allclose(a, b, atol=1e-5)
    tg, to = net.siamese_net.test_on_batch([L[100:150], R[100:150]], y[100:150]), o.test_on_batch([L[100:150], R[100:150]], y[100:150])
    np.testing.assert_allclose(tg, to, rtol=2e-5, atol=2e-6)
    np.random.seed(4)
    hg = net.finetune([L, R], y, 2, 16, verbose=0)
    np.random.seed(4)
    ho = o.fit([L, R], y, batch_size=16, epochs=2, validation_split=0.2)
    for k in ("loss", "acc", "val_loss", "val_acc"):
        np.testing.assert_allclose(hg[k], ho[k], rtol=2e-4, atol=2e-5)


def test_sigmoid_head_equals_softmax_head_with_zero_column(gpu):
    """sigmoid(z) == softmax([0, z])[1]: the two kernels' forward paths agree."""
    from a_link_amd import siamese, siamese3
    a = siamese3.SiameseNetwork((512,), "a", 0.1, seed=5)
    b = siamese.SiameseNetwork((512,), "b", 0.1, seed=6)
    ws = a.siamese_net.get_weights()
    W3 = np.concatenate([np.zeros((64, 1), np.float32), ws[4]], axis=1)
    b.siamese_net.set_weights(ws[:4] + [W3, np.array([0, ws[5][0]], np.float32)])
    L, R = _data(1000, 512, 2)
    np.testing.assert_allclose(a.predict([L, R])[:, 0], b.predict([L, R])[:, 1], atol=2e-7)


def test_save_load_and_score_matrix_single_output(gpu, tmp_path):
    from a_link_amd import evaluation as E, siamese3
    a = siamese3.SiameseNetwork((512,), str(tmp_path / "s3"), 0.1, seed=1)
    b = siamese3.SiameseNetwork((512,), str(tmp_path / "s3"), 0.1, seed=2)
    a.save()
    assert b.maybeLoadFromMemory()
    L, R = _data(20, 512, 3)
    assert np.array_equal(a.predict([L, R]), b.predict([L, R]))
    f = L / np.linalg.norm(L, axis=1, keepdims=True)
    S = E.score_matrix(a, f, col=0)
    assert np.array_equal(S[3], a.predict([np.repeat(f[3][None], 20, 0), f])[:, 0])


def test_baseline_active_learning_loop(gpu, tmp_path):
    from a_link_amd import existing_al, pairs, siamese3
    rng = np.random.RandomState(0)
    feats = [rng.randn(3, 512).astype(np.float32) + 2 * i for i in range(5)]
    imps = [rng.randn(3, 512).astype(np.float32) - 3 for _ in range(6)]
    model = siamese3.SiameseNetwork((512,), str(tmp_path / "base"), 0.1, seed=7)
    w0 = model.siamese_net.get_weights()
    for strategy in ("uncertainty_sampling", "margin_sampling", "entropy_sampling"):
        gen = pairs.getGenerator(pairs.getNormalGenerator(feats, 16, infinite=False),
                                 pairs.getNormalGenerator(imps, 16, infinite=False),
                                 pairs.getImposterGenerator(feats, imps, 16, infinite=False), 4 * 16, 0)
        np.random.seed(1)
        learner, n = existing_al.run_baseline(model, gen, strategy, active_ratio=0.5, out_model=str(tmp_path / "out"),
                                              verbose=0)
        assert n >= 1
        assert learner.estimator.classes_.tolist() == [0, 1]
        assert len(learner.X_training[0]) == len(learner.y_training) > 0
    assert (tmp_path / "out.h5").exists()
    assert any(not np.array_equal(a, b) for a, b in zip(w0, model.siamese_net.get_weights()))
    L, R = _data(10, 512, 9)
    pp = learner.predict_proba([L, R])
    assert pp.shape == (10, 2) and np.allclose(pp.sum(1), 1.0, atol=1e-6)      # hstack([1 - p, p])
    assert set(np.unique(learner.predict([L, R])).tolist()) <= {0, 1}
    assert 0.0 <= learner.score([L, R], np.zeros((10, 1), int)) <= 1.0

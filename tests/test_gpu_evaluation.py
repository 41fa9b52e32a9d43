"""GPU: N x N verification score matrix (alink_pair_scores_matrix) and DFW protocol counts
(alink_roc_counts) against the oracle's restatement of utilities/generateMatrixDFW.py and
utilities/ROC_precompute.py; full-size (7771 x 7771) run against the golden TPR/FPR the reference
script itself produced."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _feats(n, d, seed):
    rng = np.random.RandomState(seed)
    f = rng.randn(n, d).astype(np.float32)
    return f / np.linalg.norm(f, axis=1, keepdims=True)


@pytest.mark.parametrize("n,d,col", [(1, 512, 0), (37, 512, 0), (130, 2048, 0), (65, 512, 1)])
def test_score_matrix_matches_oracle(gpu, n, d, col):
    from a_link_amd import evaluation as E, siamese
    from oracle import evaluation as OE, siamese_head as O
    net = siamese.SiameseNetwork((d,), "m", 0.1, seed=n)
    ws = net.siamese_net.get_weights()
    f = _feats(n, d, n + 1)
    got = E.score_matrix(net, f, col=col, rows_per_call=16)
    want = OE.score_matrix(lambda X: O.forward(ws, X[0], X[1]), f, col=col)
    assert got.shape == (n, n)
    np.testing.assert_allclose(got, want, atol=3e-6)
    # the matrix is what predict() gives pair by pair (generateMatrixDFW.py:33 keeps out[0])
    i = n // 2
    row = net.predict([np.repeat(f[i][None], n, 0), f])[:, col]
    assert np.array_equal(got[i], row)


def test_score_matrix_committee_mean(gpu):
    from a_link_amd import committee, evaluation as E, siamese
    f = _feats(50, 512, 3)
    members = [siamese.SiameseNetwork((512,), "c%d" % i, 0.1, seed=20 + i) for i in range(3)]
    bag = committee.Bagging(members, [])
    got = E.score_matrix(bag, f, col=0)
    li = np.repeat(np.arange(50, dtype=np.int32), 50)
    ri = np.tile(np.arange(50, dtype=np.int32), 50)
    want = bag.predict_indexed(f, f, li, ri).cpu().numpy()[:, 0].reshape(50, 50)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,T", [(2, 1), (300, 25), (1031, 1000), (500, 9000)])
def test_roc_counts_match_oracle(gpu, n, T):
    from a_link_amd import evaluation as E
    from oracle import evaluation as OE
    rng = np.random.RandomState(n)
    s = rng.rand(n, n).astype(np.float32)
    s[rng.rand(n, n) < 0.05] = 0.5                      # ties with a threshold
    m = rng.randint(0, 6, (n, n))                       # code 5 = unknown -> unused
    if n == 2:
        m[0, 1] = 1
    thr = rng.rand(T)
    thr[0] = 0.5
    for case in (1, 2, 3):
        gen, imp = OE.genuine_impostor(s, m, case)
        if len(gen) == 0 or len(imp) == 0:
            with pytest.raises(ZeroDivisionError):
                E.roc_precompute(s, m, thr, case)
            continue
        tpr, fpr = E.roc_precompute(s, m, thr, case)
        wt, wf = OE.roc_precompute(s, m, thr, case)
        assert np.array_equal(tpr, wt) and np.array_equal(fpr, wf)
    with pytest.raises(ValueError):
        E.roc_precompute(s, m, thr, 4)


def test_roc_full_size_against_reference_output(gpu):
    """7771 x 7771 (the size utilities/ROC_precompute.py hard-codes): inputs regenerated from the
    fixture's seed, TPR/FPR compared with what the reference script wrote."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mge", os.path.join(GOLD, "make_golden_eval.py"))
    mge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mge)
    from a_link_amd import evaluation as E
    with np.load(os.path.join(GOLD, "eval_roc.npz")) as z:
        scores, mask, thr = mge.eval_inputs(int(z["seed"]))
        assert np.array_equal(thr, z["thresholds"])
        sd = torch.from_numpy(scores.astype(np.float32)).cuda()
        assert np.array_equal(sd.cpu().numpy().astype(np.float64), scores.astype(np.float32).astype(np.float64))
        md = torch.from_numpy(mask.astype(np.uint8)).cuda()
        for case in (1, 2, 3):
            tpr, fpr = E.roc_precompute(sd, md, thr, case)
            np.testing.assert_allclose(tpr, z["case%d" % case][0], rtol=0, atol=1e-15)
            np.testing.assert_allclose(fpr, z["case%d" % case][1], rtol=0, atol=1e-15)

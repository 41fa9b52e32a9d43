"""CPU: A-LINK's query-selection rule (reference code/ALINK_arc.py:167-198, code/ALINK.py:170-201):
hand-made known-answer cases + product (selection.py) == oracle (al_logic.py) on random inputs."""
import numpy as np

import a_link_amd  # noqa: F401
from a_link_amd import helpers, selection
from oracle import al_logic as O


def _probs(c0):
    c0 = np.asarray(c0, np.float32)
    return np.stack([c0, 1 - c0], axis=1)


def test_known_answer_column0():
    # 8 pairs, 2 noises, ratio .5 -> 4 most-disparate per noise
    ens = _probs([0.90, 0.10, 0.52, 0.80, 0.30, 0.95, 0.05, 0.60])
    n1 = _probs([0.10, 0.15, 0.10, 0.79, 0.90, 0.20, 0.06, 0.61])     # |d| = .8 .05 .42 .01 .6 .75 .01 .01
    n2 = _probs([0.20, 0.90, 0.12, 0.10, 0.35, 0.15, 0.90, 0.59])     # |d| = .7 .8 .4 .7 .05 .8 .85 .01
    y = np.array([[1], [0], [1], [1], [0], [0], [0], [1]])
    # noise1 top-4: {0,5,4,2}; noise2 top-4: {6,1,5,(0|3 tie at .7 -> lower index 0)} -> both: {0,5}
    q, active, labels = selection.select_queries(ens, [n1, n2], y, col=0, disparity_ratio=0.5, eps=0.05)
    # ens[0][0]=.9 -> c1 True, y=1 -> keep; ens[5][0]=.95 -> True but y=0 -> oracle disagrees, dropped
    assert q == [0] and active == 2
    assert np.array_equal(labels, [[1]])
    qs, act = O.select_queries(ens, [n1, n2], y, 0, 0.5, 0.05)
    assert qs == {0} and act == 2


def test_grey_band_and_blind_strategy():
    ens = _probs([0.52, 0.48, 0.56, 0.44, 0.9])
    n1 = _probs([0.10, 0.90, 0.10, 0.90, 0.1])
    y = np.array([[1], [0], [1], [0], [0]])
    q, active, _ = selection.select_queries(ens, [n1], y, col=0, disparity_ratio=1.0, eps=0.05)
    assert q == [2, 3] and active == 3           # .52/.48 inside (0.45,0.55); pair 4 queried but wrong
    q, active, _ = selection.select_queries(ens, [n1], y, col=0, disparity_ratio=0.25, eps=0.05, blind_strategy=True)
    assert q == [2, 3] and active == 3           # blind: decision flips on all 5
    assert selection.partition_by_noise([5, 6, 7, 8, 9], 2) == [[5, 6], [7, 8]]


def test_roundoff_and_column_conventions():
    assert np.array_equal(helpers.roundoff([0.5, 0.49, 1.0]), [[1], [0], [1]])
    assert np.array_equal(O.roundoff([0.5, 0.49, 1.0]), [[1], [0], [1]])


def test_random_inputs_product_equals_oracle():
    rng = np.random.RandomState(0)
    for trial in range(20):
        P = rng.randint(5, 400)
        nn = rng.randint(1, 5)
        ens = _probs(rng.rand(P))
        dis = [_probs(rng.rand(P)) for _ in range(nn)]
        y = rng.randint(0, 2, (P, 1))
        col = trial % 2
        ratio = [0.25, 0.5, 1.0][trial % 3]
        for blind in (False, True):
            q, a, lab = selection.select_queries(ens, dis, y, col=col, disparity_ratio=ratio, eps=0.05,
                                                 blind_strategy=blind)
            qs, ao = O.select_queries(ens, dis, y, col, ratio, 0.05, blind)
            assert set(q) == qs and a == ao and q == sorted(q)
            assert len(lab) == len(q)

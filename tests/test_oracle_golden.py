"""CPU: the oracle's AL-logic restatement and the product's host logic both reproduce the golden
vectors captured from the reference's own functions (tests/golden/make_golden.py)."""
import os

import numpy as np

import a_link_amd  # noqa: F401
from a_link_amd import pairs, uncertainty as U
from oracle import al_logic as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(G, name)) as z:
        return {k: z[k] for k in z.files}


def test_uncertainty_measures_bit_exact():
    g = _load("uncertainty.npz")
    for tag in ("p2", "p5"):
        p = g[tag]
        for fn_o, fn_p, key in ((O.proba_uncertainty, U._proba_uncertainty, "_uncertainty"),
                                (O.proba_margin, U._proba_margin, "_margin"),
                                (O.proba_entropy, U._proba_entropy, "_entropy")):
            want = g[tag + key]
            for fn in (fn_o, fn_p):
                got = fn(p)
                assert got.dtype == want.dtype and got.shape == want.shape
                assert np.array_equal(got, want), (tag, key)
    # the survey's sanity values
    np.testing.assert_allclose(g["p2_uncertainty"][:3], [0.1, 0.5, 0.3], atol=1e-6)
    np.testing.assert_allclose(g["p2_margin"][:3], [0.8, 0.0, 0.4], atol=1e-6)
    np.testing.assert_allclose(g["p2_entropy"][:3], [0.32508297, 0.69314718, 0.6108643], atol=1e-6)


def test_samplers_match_reference_sets_and_left_twice_quirk():
    g = _load("uncertainty.npz")
    p = g["p2"]

    class Clf(object):
        def predict_proba(self, X, **kw):
            return p
    X = [np.arange(257)[:, None].astype(np.float32), np.arange(257)[:, None].astype(np.float32) + 1000]
    for name in ("uncertainty_sampling", "margin_sampling", "entropy_sampling"):
        idx, inst = getattr(U, name)(Clf(), X, n_instances=20)
        assert set(idx.tolist()) == set(g[name + "_idx20"].tolist())
        assert np.array_equal(inst[0], inst[1])                       # left twice (code/uncertainty.py:159)
        assert np.array_equal(np.sort(inst[0].ravel()), np.sort(g[name + "_inst0"].ravel()))
        assert np.array_equal(g[name + "_inst0"], g[name + "_inst1"])


def test_bagging_mean_bit_exact():
    g = _load("bagging.npz")
    mem = g["members"]
    assert np.array_equal(O.bagging_predict(list(mem)), g["mean3"])
    assert np.array_equal(O.bagging_predict(list(mem[:2])), g["mean2"])
    assert np.array_equal(O.bagging_predict([mem[0]]), g["mean1"])
    from a_link_amd import committee

    class Member(object):
        def __init__(self, p):
            self.p = p

        def predict(self, X):
            return self.p
    assert np.array_equal(committee.Bagging([Member(m) for m in mem], []).predict(None), g["mean3"])
    # sequential f32 accumulate then one divide — the order the HIP committee kernel uses
    seq = ((mem[0] + mem[1]) + mem[2]) / np.float32(3)
    assert np.array_equal(seq, g["mean3"])


def test_create_minibatch_order_and_labels():
    g = _load("minibatch.npz")
    n_plain, n_dig = g["n_plain"].tolist(), g["n_dig"].tolist()
    code, plain, dig = 0, [], []
    for k in n_plain:
        plain.append(np.arange(code, code + k, dtype=np.float32).reshape(k, 1, 1, 1)); code += k
    for k in n_dig:
        dig.append(np.arange(code, code + k, dtype=np.float32).reshape(k, 1, 1, 1)); code += k
    for fn in (O.create_minibatch, pairs.createMiniBatch):
        (xl, xr), y = fn(plain, dig)
        assert np.array_equal(xl.ravel().astype(np.int64), g["left_ids"])
        assert np.array_equal(xr.ravel().astype(np.int64), g["right_ids"])
        assert np.array_equal(y, g["y"])
    li, ri, y = pairs.createMiniBatchIndices(n_plain, n_dig)
    assert np.array_equal(li, g["left_ids"]) and np.array_equal(ri, g["right_ids"]) and np.array_equal(y, g["y"])
    P = sum(n_plain) * sum(n_dig) + sum(n_dig) ** 2
    assert len(li) == P
    pre, post = pairs.splitDisguiseData(dig, 0.5)
    assert [len(a) for a in pre] == g["split_pre_len"].tolist()
    assert [a.ravel()[0] for a in post] == g["split_post_first"].tolist()

"""CPU: the oracle's restatement of the perturbation stage, the batched differential evolution, the
few-pixel attack and the DFW evaluation utilities against golden vectors recorded from the
reference's own code (tests/golden/make_golden_noise.py, make_golden_eval.py)."""
import os

import numpy as np
import pytest

from oracle import de as ODE
from oracle import evaluation as OE
from oracle import noise as ON

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gn():
    with np.load(os.path.join(GOLD, "noise.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def gd():
    with np.load(os.path.join(GOLD, "de.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name,fn", [("gaussian", ON.ref_gaussian), ("speckle", ON.ref_speckle),
                                     ("poisson", ON.ref_poisson)])
def test_noise_matches_reference_stream(gn, name, fn):
    for tag in ("", "_f"):
        np.random.seed(1234)
        got = fn(gn["img" + tag])
        assert got.dtype == gn[name + tag].dtype
        assert np.array_equal(got, gn[name + tag]), name + tag


def test_pair_noise_order(gn):
    pair = [np.stack([gn["img"], gn["img_f"]]), np.stack([gn["img_f"], gn["img"]])]
    np.random.seed(77)
    left = np.array([ON.ref_gaussian(im) for im in pair[0]])       # left half first (code/noise.py:27-28)
    right = np.array([ON.ref_gaussian(im) for im in pair[1]])
    assert np.array_equal(left, gn["pair_left"]) and np.array_equal(right, gn["pair_right"])


@pytest.mark.parametrize("size", [224, 150])
def test_perlin_matches_reference(gn, size):
    np.random.seed(4321)
    z = ON.ref_perlin(np.zeros((size, size, 3)))
    assert np.array_equal(z[..., 0], z[..., 2])
    np.testing.assert_allclose(z[::7, ::5, 0], gn["perlin%d_sub" % size], rtol=0, atol=1e-10)
    m = np.array([z[..., 0].sum(), (z[..., 0] ** 2).sum(), z[..., 0].min(), z[..., 0].max()])
    np.testing.assert_allclose(m, gn["perlin%d_moments" % size], rtol=1e-10, atol=1e-8)


def test_perlin_112_raises_like_reference(gn):
    assert int(gn["perlin112_raises"]) == 1
    with pytest.raises(ValueError):
        ON.ref_perlin(np.zeros((112, 112, 3)))


def test_perturb_image(gn):
    assert np.array_equal(ON.perturb_image(gn["perturb_xs"], gn["img"]), gn["perturb_out"])
    assert np.array_equal(ON.perturb_image(gn["perturb_xs"][0], gn["img"]), gn["perturb_one"])


def test_saltpepper_tuple_semantics_known_answer():
    img = np.full((20, 30, 3), 7.0, np.float32)
    np.random.seed(3)
    out = ON.ref_saltpepper(img)
    n_salt, n_pepper = ON.salt_pepper_counts(img.shape)
    assert (n_salt, n_pepper) == (4, 4)                      # ceil(0.004 * 1800 * 0.5)
    changed = np.argwhere(out != 7.0)
    assert 1 <= len(changed) <= 8
    assert changed[:, 0].max() <= 18 and changed[:, 1].max() <= 28 and changed[:, 2].max() <= 1   # randint(0, i-1)
    assert set(np.unique(out[out != 7.0])) <= {0.0, 1.0}


def _rosen(xs):
    xs = np.atleast_2d(xs)
    return (100.0 * (xs[:, 1:] - xs[:, :-1] ** 2) ** 2 + (1 - xs[:, :-1]) ** 2).sum(axis=1)


DE_CASES = {"best1bin": dict(popsize=5, maxiter=12, seed=lambda: np.random.RandomState(3)),
            "rand1exp": dict(strategy="rand1exp", popsize=4, maxiter=8, seed=lambda: 11, mutation=0.7),
            "best2bin_tol": dict(strategy="best2bin", popsize=6, maxiter=200, tol=0.5, seed=lambda: 2),
            "attack_like": dict(popsize=1, maxiter=9, recombination=1, atol=-1, seed=lambda: np.random.RandomState(9)),
            "currenttobest1bin": dict(strategy="currenttobest1bin", popsize=3, maxiter=6, seed=lambda: 4),
            "randtobest1exp": dict(strategy="randtobest1exp", popsize=3, maxiter=6, seed=lambda: 8, recombination=0.9)}


@pytest.mark.parametrize("name", sorted(DE_CASES))
def test_de_retraces_reference(gd, name):
    kw = dict(DE_CASES[name])
    kw["seed"] = kw["seed"]()
    with np.errstate(divide="ignore"):
        r = ODE.differential_evolution(_rosen, [(-2, 2)] * 4, **kw)
    assert np.array_equal(r.x, gd[name + "_x"])
    assert r.fun == gd[name + "_fun"]
    assert [r.nit, r.nfev] == list(gd[name + "_nit_nfev"])


def test_de_callback_stop(gd):
    r = ODE.differential_evolution(_rosen, [(-2, 2)] * 4, popsize=5, maxiter=50, seed=21,
                                   callback=lambda x, convergence: bool(_rosen(x)[0] < 5.0))
    assert np.array_equal(r.x, gd["callback_x"]) and [r.nit, r.nfev] == list(gd["callback_nit_nfev"])
    assert not r.success


def test_pixel_attack_retraces_reference(gd):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgn", os.path.join(GOLD, "make_golden_noise.py"))
    mgn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgn)
    pimg = list(gd["attack_in"])
    model = mgn.ToyPairModel(pimg[0].shape, 0)
    np.random.seed(99)
    with np.errstate(divide="ignore"):
        res = ODE.PixelAttacker(model).attack_all(pimg, [[1, 0], [0, 1]], dimensions=(24, 12), pixel_count=3,
                                                  maxiter=6, popsize=30)
    assert np.array_equal(np.stack(res), gd["attack_out"])


# ---- evaluation ---------------------------------------------------------------------------------------
def test_get_stats_matches_reference_prints():
    with np.load(os.path.join(GOLD, "eval_stats.npz")) as z:
        for i in range(4):
            tpr, fpr = z["curve%d" % i]
            got = OE.get_stats(tpr, fpr)
            np.testing.assert_allclose(got, z["stats"][i], rtol=0, atol=5.1e-7)     # printed with %f


def test_roc_precompute_small_known_answer():
    s = np.array([[0, .9, .2, .6], [0, 0, .5, .5], [0, 0, 0, .1], [0, 0, 0, 0]])
    m = np.array([[0, 1, 3, 2], [0, 0, 4, 1], [0, 0, 0, 3], [0, 0, 0, 0]])
    tpr, fpr = OE.roc_precompute(s, m, [0.5, 0.95, 0.0], 3)
    assert np.allclose(tpr, [1.0, 0.0, 1.0]) and np.allclose(fpr, [1 / 3, 0.0, 1.0])
    tpr, fpr = OE.roc_precompute(s, m, [0.5], 1)
    assert np.allclose(tpr, [1.0]) and np.allclose(fpr, [0.0])


def test_roc_precompute_full_size_matches_reference_output():
    """7771 x 7771 as hard-coded by the reference script; inputs regenerated from the fixture's seed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mge", os.path.join(GOLD, "make_golden_eval.py"))
    mge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mge)
    with np.load(os.path.join(GOLD, "eval_roc.npz")) as z:
        scores, mask, thr = mge.eval_inputs(int(z["seed"]))
        for case in (1, 2, 3):
            tpr, fpr = OE.roc_precompute(scores, mask, thr, case)
            np.testing.assert_allclose(tpr, z["case%d" % case][0], rtol=0, atol=1e-15)
            np.testing.assert_allclose(fpr, z["case%d" % case][1], rtol=0, atol=1e-15)


def test_float32_ptrs_sampler_draws_poisson():
    """The float32 PTRS sampler the device and the oracle share since round 5 (oracle/noise.py::_ptrs_f32 = noise.hip::
    poisson_ptrs_f32: Hörmann's transformed rejection with the pmf written without cancellation, a table of log k! below 8)
    must DRAW Poisson(lam) — the reference's np.random.poisson(image * vals) (code/noise.py:75) fixes the distribution, not the
    stream.  Chi-square against scipy.stats.poisson at the edge of the method's range (lam = 10: small-k table), a
    fractional lam, the image scale (x * 256) and the resized-image scale (x * 65536); mean within 4 standard errors."""
    from scipy import stats
    from oracle import noise as ON
    n = 120000
    for lam, seed in ((10.0, 1), (12.3, 2), (177.0 * 256.0, 3), (200.5 * 65536.0, 4)):
        L = np.full(n, lam, np.float32)
        k = ON._ptrs_f32(L, np.arange(n, dtype=np.uint64) + np.uint64(seed * 10 ** 7), seed)
        lam = float(L[0])
        assert abs(k.mean() - lam) < 4.0 * np.sqrt(lam / n), (lam, k.mean())
        assert abs(k.var() / lam - 1.0) < 0.03, (lam, k.var() / lam)
        lo, hi = int(stats.poisson.ppf(1e-4, lam)), int(stats.poisson.ppf(1 - 1e-4, lam))
        cuts = np.unique(np.linspace(lo, hi, min(60, hi - lo + 1)).astype(np.int64))      # bins (-inf, c0], (c0, c1], ..., (c_last, inf)
        below = np.array([(k <= c).sum() for c in cuts], float)
        obs = np.diff(np.concatenate([[0.0], below, [float(n)]]))
        exp = np.diff(np.concatenate([[0.0], stats.poisson.cdf(cuts, lam), [1.0]])) * n
        m = exp > 5
        chi = ((obs[m] - exp[m]) ** 2 / exp[m]).sum()
        assert stats.chi2.sf(chi, int(m.sum()) - 1) > 1e-4, (lam, chi, int(m.sum()))

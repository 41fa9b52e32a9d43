#!/usr/bin/env python3
"""Generates tests/golden/noise.npz and de.npz by IMPORTING the reference's own code/noise.py,
code/attack.py and code/differential_evolution.py (this container only) and recording their outputs
on seeded inputs.  Only data is stored.

Absent third-party modules get EMPTY placeholders (cv2, keras.*), as in make_golden.py.  SciPy 1.15
no longer has two private names the vendored solver imports (code/differential_evolution.py:16,18):
`scipy.optimize.optimize._status_message` (a dict of three message strings) and `scipy._lib.six`
(`xrange`, `string_types`); they are provided as placeholders too — they hold no arithmetic.

SaltPepper is not recorded: with this container's NumPy the reference's list index means something
else than it did under the NumPy the reference was written for.

Run:  python tests/golden/make_golden_noise.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/code"
HERE = os.path.dirname(os.path.abspath(__file__))


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_placeholders():
    _placeholder("cv2")
    k = _placeholder("keras")
    k.models = _placeholder("keras.models", Model=None, Sequential=None, clone_model=None)
    k.layers = _placeholder("keras.layers", Input=None, Lambda=None, Activation=None)
    k.backend = _placeholder("keras.backend")
    import scipy.optimize
    import scipy._lib
    so = _placeholder("scipy.optimize.optimize", _status_message={
        'success': 'Optimization terminated successfully.',
        'maxfev': 'Maximum number of function evaluations has been exceeded.',
        'maxiter': 'Maximum number of iterations has been exceeded.'})
    scipy.optimize.optimize = so
    six = _placeholder("scipy._lib.six", xrange=range, string_types=(str,))
    scipy._lib.six = six


class ToyPairModel(object):
    """Deterministic stand-in for noise.PredictionWrappedModel: two-class 'probabilities' from a fixed
    random projection of the stacked pair image (the reference only needs .predict -> (n, 2))."""

    def __init__(self, shape, seed):
        r = np.random.RandomState(seed)
        self.w = r.randn(int(np.prod(shape))) / 255.0 / np.sqrt(np.prod(shape))
        self.b = 0.3

    def predict(self, X):
        X = np.asarray(X, dtype=np.float64)
        z = X.reshape(len(X), -1) @ self.w * 40.0 + self.b
        p1 = 1.0 / (1.0 + np.exp(-z))
        return np.stack([1 - p1, p1], axis=1).astype(np.float32)


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    import noise as ref_noise                       # reference code/noise.py
    import attack as ref_attack                     # reference code/attack.py
    from differential_evolution import differential_evolution as ref_de

    out = {}
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (12, 12, 3)).astype(np.float32)
    img_f = (img * 0.37 + 1.25).astype(np.float32)            # non-integer pixels (bilinear-resized faces)
    out["img"], out["img_f"] = img, img_f
    for name, cls in (("gaussian", ref_noise.Gaussian), ("speckle", ref_noise.Speckle), ("poisson", ref_noise.Poisson)):
        for tag, im in (("", img), ("_f", img_f)):
            np.random.seed(1234)
            out[name + tag] = cls().addIndividualNoise(im)
    # addPairNoise draws left images first, then right (code/noise.py:25-30)
    pair = [np.stack([img, img_f]), np.stack([img_f, img])]
    np.random.seed(77)
    pl, pr = ref_noise.Gaussian().addPairNoise(pair, [0, 1])
    out["pair_left"], out["pair_right"] = pl, pr
    # Perlin at the two octave sets; stored as a strided subsample + moments (the full field is 50k values)
    for size in (224, 150):
        np.random.seed(4321)
        z = ref_noise.Perlin().addIndividualNoise(np.zeros((size, size, 3)))
        assert np.array_equal(z[..., 0], z[..., 1]) and np.array_equal(z[..., 0], z[..., 2])
        out["perlin%d_sub" % size] = z[::7, ::5, 0]
        out["perlin%d_moments" % size] = np.array([z[..., 0].sum(), (z[..., 0] ** 2).sum(), z[..., 0].min(), z[..., 0].max()])
    try:
        np.random.seed(1)
        ref_noise.Perlin().addIndividualNoise(np.zeros((112, 112, 3)))
        out["perlin112_raises"] = np.array(0)
    except ValueError:
        out["perlin112_raises"] = np.array(1)
    # perturb_image (code/attack.py:5-29)
    xs = np.array([[1.9, 2.2, 255.7, 0.1, 17.0, 11, 11, 1, 2, 3],
                   [0, 0, 9, 8, 7, 0.5, 0.99, 100, 101, 102.9]])
    out["perturb_xs"] = xs
    out["perturb_out"] = ref_attack.perturb_image(xs, img)
    out["perturb_one"] = ref_attack.perturb_image(xs[0], img)
    np.savez_compressed(os.path.join(HERE, "noise.npz"), **out)

    # ---- the batched differential evolution (code/differential_evolution.py) -----------------------
    de = {}

    def rosen(xs):                                   # population-at-once objective
        xs = np.atleast_2d(xs)
        return (100.0 * (xs[:, 1:] - xs[:, :-1] ** 2) ** 2 + (1 - xs[:, :-1]) ** 2).sum(axis=1)
    bounds = [(-2, 2)] * 4
    cases = {"best1bin": dict(popsize=5, maxiter=12, seed=np.random.RandomState(3)),
             "rand1exp": dict(strategy="rand1exp", popsize=4, maxiter=8, seed=11, mutation=0.7),
             "best2bin_tol": dict(strategy="best2bin", popsize=6, maxiter=200, tol=0.5, seed=2),
             "attack_like": dict(popsize=1, maxiter=9, recombination=1, atol=-1, seed=np.random.RandomState(9)),
             "currenttobest1bin": dict(strategy="currenttobest1bin", popsize=3, maxiter=6, seed=4),
             "randtobest1exp": dict(strategy="randtobest1exp", popsize=3, maxiter=6, seed=8, recombination=0.9)}
    for name, kw in cases.items():
        kw = dict(kw)
        r = ref_de(rosen, bounds, polish=False, **kw)
        de[name + "_x"], de[name + "_fun"] = r.x, np.array(r.fun)
        de[name + "_nit_nfev"] = np.array([r.nit, r.nfev])
    # early stop from the callback
    r = ref_de(rosen, bounds, polish=False, popsize=5, maxiter=50, seed=21,
               callback=lambda x, convergence: bool(rosen(x)[0] < 5.0))
    de["callback_x"], de["callback_nit_nfev"] = r.x, np.array([r.nit, r.nfev])

    # ---- PixelAttacker.attack_all on a toy pair model (code/attack.py:91-103) -----------------------
    pimg = [np.concatenate([img, img_f], axis=0), np.concatenate([img_f, img], axis=0)]     # (24, 12, 3) stacked pairs
    model = ToyPairModel(pimg[0].shape, 0)
    np.random.seed(99)
    res = ref_attack.PixelAttacker(model).attack_all(pimg, [[1, 0], [0, 1]], dimensions=(24, 12), pixel_count=3,
                                                     maxiter=6, popsize=30)
    de["attack_in"] = np.stack(pimg)
    de["attack_out"] = np.stack(res)
    de["attack_probs_before"] = model.predict(np.stack(pimg))
    de["attack_probs_after"] = model.predict(np.stack(res))
    np.savez_compressed(os.path.join(HERE, "de.npz"), **de)
    print("wrote noise.npz, de.npz")


if __name__ == "__main__":
    main()

"""CPU: the product's host-side differential evolution (a-link_amd/differential_evolution.py).
rng_compat=True must retrace the trajectories recorded from the reference's own solver
(tests/golden/de.npz); the vectorised default must draw valid, distinct samples and converge."""
import os

import numpy as np
import pytest

import a_link_amd  # noqa: F401
from a_link_amd.differential_evolution import DifferentialEvolutionSolver, differential_evolution

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rosen(xs):
    xs = np.atleast_2d(xs)
    return (100.0 * (xs[:, 1:] - xs[:, :-1] ** 2) ** 2 + (1 - xs[:, :-1]) ** 2).sum(axis=1)


CASES = {"best1bin": dict(popsize=5, maxiter=12, seed=lambda: np.random.RandomState(3)),
         "rand1exp": dict(strategy="rand1exp", popsize=4, maxiter=8, seed=lambda: 11, mutation=0.7),
         "best2bin_tol": dict(strategy="best2bin", popsize=6, maxiter=200, tol=0.5, seed=lambda: 2),
         "attack_like": dict(popsize=1, maxiter=9, recombination=1, atol=-1, seed=lambda: np.random.RandomState(9)),
         "currenttobest1bin": dict(strategy="currenttobest1bin", popsize=3, maxiter=6, seed=lambda: 4),
         "randtobest1exp": dict(strategy="randtobest1exp", popsize=3, maxiter=6, seed=lambda: 8, recombination=0.9)}


@pytest.mark.parametrize("name", sorted(CASES))
def test_compat_mode_retraces_reference(name):
    with np.load(os.path.join(GOLD, "de.npz")) as gd:
        kw = dict(CASES[name])
        kw["seed"] = kw["seed"]()
        r = differential_evolution(_rosen, [(-2, 2)] * 4, polish=False, rng_compat=True, **kw)
        assert np.array_equal(r.x, gd[name + "_x"]) and r.fun == gd[name + "_fun"]
        assert [r.nit, r.nfev] == list(gd[name + "_nit_nfev"])


def test_compat_callback_stop():
    with np.load(os.path.join(GOLD, "de.npz")) as gd:
        r = differential_evolution(_rosen, [(-2, 2)] * 4, polish=False, popsize=5, maxiter=50, seed=21, rng_compat=True,
                                   callback=lambda x, convergence: bool(_rosen(x)[0] < 5.0))
        assert np.array_equal(r.x, gd["callback_x"]) and [r.nit, r.nfev] == list(gd["callback_nit_nfev"])
        assert r.success is False and "callback" in r.message


@pytest.mark.parametrize("strategy", ["best1bin", "rand1exp", "best2bin", "rand2bin", "currenttobest1exp",
                                      "randtobest1bin"])
def test_fast_mode_samples_and_convergence(strategy):
    s = DifferentialEvolutionSolver(_rosen, [(-2, 2)] * 4, strategy=strategy, popsize=6, seed=5, polish=False,
                                    maxiter=400, tol=1e-8)
    s.scale = 0.7                                      # __next__ sets the dithered scale before drawing trials
    trials = s._trials_fast(s.num_population_members)
    assert trials.shape == s.population.shape and trials.min() >= 0 and trials.max() <= 1
    r = s.solve()
    assert r.fun < 1e-4, (strategy, r.fun)


def test_fast_mode_sample_indices_are_distinct_and_exclude_candidate():
    s = DifferentialEvolutionSolver(_rosen, [(-2, 2)] * 3, strategy="rand2bin", popsize=2, seed=0, polish=False)
    npop = s.num_population_members                    # 6: every draw is nearly a permutation
    seen = np.zeros((npop, npop), int)
    orig = s._base_and_diff

    def spy(kind, cand, idx):
        for c, row in zip(cand, idx):
            assert len(set(row.tolist())) == len(row) and c not in row and row.min() >= 0 and row.max() < npop
            seen[c, row] += 1
        return orig(kind, cand, idx)
    s._base_and_diff = spy
    s.scale = 0.7
    for _ in range(200):
        s._trials_fast(npop)
    off = seen[~np.eye(npop, dtype=bool)]
    assert off.min() > 0.7 * off.mean()                # uniform over the other members


def test_errors_like_scipy():
    with pytest.raises(ValueError):
        differential_evolution(_rosen, [(-2, 2)] * 4, strategy="nope")
    with pytest.raises(ValueError):
        differential_evolution(_rosen, [(-2, 2)] * 4, mutation=2.5)
    with pytest.raises(ValueError):
        differential_evolution(_rosen, [(-2, np.inf)] * 4)


def test_polish_improves_or_keeps():
    r0 = differential_evolution(_rosen, [(-2, 2)] * 4, polish=False, seed=1, maxiter=30)
    r1 = differential_evolution(_rosen, [(-2, 2)] * 4, polish=True, seed=1, maxiter=30)
    assert r1.fun <= r0.fun

"""CPU: the in-memory pair generators (a-link_amd/pairs.py) against batches recorded from the reference's
own generators (tests/golden/make_golden_generators.py imports code/readDFW3.py), and the drivers' flags."""
import importlib.util
import os

import numpy as np
import pytest

import a_link_amd  # noqa: F401
from a_link_amd import pairs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _people():
    spec = importlib.util.spec_from_file_location("mgg", os.path.join(GOLD, "make_golden_generators.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.people(5, 1), m.people(4, 2)


def test_generators_match_reference_batches():
    plain, imp = _people()
    with np.load(os.path.join(GOLD, "generators.npz")) as g:
        gen = pairs.getNormalGenerator(plain, 4, infinite=False)
        for i in range(6):
            (xl, xr), y = next(gen)
            assert np.array_equal(xl, g["normal_left"][i]) and np.array_equal(xr, g["normal_right"][i])
            assert np.array_equal(y, g["normal_y"][i])
        gen = pairs.getImposterGenerator(plain, imp, 5, infinite=True)
        for i in range(4):
            (xl, xr), y = next(gen)
            assert np.array_equal(xl, g["imp_left"][i]) and np.array_equal(xr, g["imp_right"][i]) and np.array_equal(y, g["imp_y"][i])
        np.random.seed(123)
        gen = pairs.getGenerator(pairs.getNormalGenerator(plain, 8), pairs.getNormalGenerator(imp, 8),
                                 pairs.getImposterGenerator(plain, imp, 8), 10)
        for i in range(3):
            (xl, xr), y = next(gen)
            assert np.array_equal(xl, g["mix%d_left" % i]) and np.array_equal(xr, g["mix%d_right" % i])
            assert np.array_equal(y, g["mix%d_y" % i])


def test_finite_generators_end():
    plain, imp = _people()
    n = sum(1 for _ in pairs.getNormalGenerator(plain, 4, infinite=False))
    total = sum(len(p) for p in plain) ** 2
    assert n == total // 4                                   # the tail shorter than a batch is dropped
    gen = pairs.getGenerator(pairs.getNormalGenerator(plain, 8, infinite=False), pairs.getNormalGenerator(imp, 8, infinite=False),
                             pairs.getImposterGenerator(plain, imp, 8, infinite=False), 10)
    np.random.seed(0)
    assert 1 <= sum(1 for _ in gen) < 50                     # ends when a source is exhausted


def test_flags_match_reference_defaults():
    import argparse
    from a_link_amd import alink_loop as AL
    f = AL.add_flags(argparse.ArgumentParser()).parse_args([])
    assert (f.ft_epochs, f.batch_size, f.batch_send, f.mixture_ratio, f.alink_bs) == (3, 16, 64, 2, 16)
    assert (f.active_ratio, f.split_ratio, f.disparity_ratio, f.eps) == (1.0, 0.5, 0.25, 0.05)
    assert f.noise == 'gaussian,saltpepper,poisson,perlin,speckle,adversarial' and f.blind_strategy is False
    with pytest.raises(AttributeError):
        AL.Flags(nope=1)


def test_createMiniBatchMTP_counts():
    from a_link_amd import alink_loop as AL
    plain, _ = _people()
    (xl, xr), y = AL.createMiniBatchMTP(plain)
    n = sum(len(p) for p in plain)
    assert len(y) == n * n and int(y.sum()) == sum(len(p) ** 2 for p in plain)

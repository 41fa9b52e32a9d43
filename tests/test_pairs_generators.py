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


def _foreign(gen):
    """the same batches through a plain generator function: mix_balanced cannot look ahead in it (the round-by-round form)"""
    for b in gen:
        yield b


def test_skipping_the_rounds_the_reference_discards_changes_nothing():
    """getGenerator (reference code/readDFW.py:180-209) draws a batch from each of its three sources per round and DROPS the
    round when the joined labels lack a class — with many persons nearly every round of the all-pairs sweep is all-negative.
    pairs.mix_balanced passes over such rounds by index arithmetic when the sources are its own; the batches it yields, and the
    np.random stream it consumes, must equal the round-by-round form's: 60 persons, infinite and finite sources, one- and
    three-source forms."""
    import time
    from a_link_amd import pairs
    rng = np.random.RandomState(3)
    people = [rng.randn(rng.randint(2, 5), 6).astype(np.float32) for _ in range(60)]
    imposters = [rng.randn(rng.randint(1, 3), 6).astype(np.float32) for _ in range(20)]

    def build(fast, infinite=True):
        srcs = [pairs.getNormalGenerator(people, 16, infinite), pairs.getNormalGenerator(people[::-1], 16, infinite),
                pairs.getImposterGenerator(people, imposters, 16, infinite)]
        if not fast:
            srcs = [_foreign(s) for s in srcs]
        return pairs.getGenerator(srcs[0], srcs[1], srcs[2], 16)

    out, took = {}, {}
    for fast in (True, False):
        np.random.seed(11)
        g = build(fast)
        t = time.perf_counter()
        out[fast] = [next(g) for _ in range(40)]
        took[fast] = time.perf_counter() - t
        out[fast].append(np.random.rand())                      # the stream afterwards is the same too
    for a, b in zip(out[True][:-1], out[False][:-1]):
        assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1]) and np.array_equal(a[1], b[1])
    assert out[True][-1] == out[False][-1]
    assert took[True] < took[False]
    # finite sources: both forms end after the same number of batches
    ends = []
    for fast in (True, False):
        np.random.seed(5)
        ends.append(len(list(build(fast, infinite=False))))
    assert ends[0] == ends[1] > 0
    # the one-source form (readMTP.getGenerator)
    one = []
    for fast in (True, False):
        np.random.seed(7)
        src = pairs.getNormalGenerator(people, 16)
        g = pairs.getGeneratorMTP(src if fast else _foreign(src), 8)
        one.append([next(g) for _ in range(10)])
    for a, b in zip(*one):
        assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[1], b[1])
    print("mix_balanced, 40 batches over 60 persons: %.1f ms skipping ahead, %.1f ms round by round" % (1e3 * took[True], 1e3 * took[False]))

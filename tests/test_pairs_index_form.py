"""CPU: pairs.BalancedMix in its index form — next_indices() must be the batch next() gathers, from the same random stream;
np.random.choice(g, m, replace=False) and g[np.random.permutation(len(g))[:m]] must be the same draws (what balance_rows
relies on, reference code/readDFW.py:189-199)."""
import numpy as np

import a_link_amd  # noqa: F401
from a_link_amd import pairs


def _feats(n, seed, d=8):
    rng = np.random.RandomState(seed)
    return [rng.randn(rng.randint(2, 6), d).astype(np.float32) for _ in range(n)]


def test_choice_without_replacement_is_a_permutation_prefix():
    for n in (1, 2, 3, 7, 16, 48, 257):
        g = np.arange(100, 100 + n)
        for m in sorted({1, n // 2 or 1, n}):
            np.random.seed(n * 1000 + m)
            a = np.random.choice(g, m, replace=False)
            sa = np.random.get_state()[1].tolist(), np.random.get_state()[2]
            np.random.seed(n * 1000 + m)
            b = g[np.random.permutation(len(g))[:m]]
            sb = np.random.get_state()[1].tolist(), np.random.get_state()[2]
            assert np.array_equal(a, b) and sa == sb, (n, m)


def test_index_form_is_the_gathered_form():
    for infinite in (True, False):
        f = _feats(30, 1)
        g1 = pairs.getGenerator(pairs.getNormalGenerator(f, 8, infinite), pairs.getNormalGenerator(f, 8, infinite),
                                pairs.getImposterGenerator(f, f, 8, infinite), 10)
        g2 = pairs.getGenerator(pairs.getNormalGenerator(f, 8, infinite), pairs.getNormalGenerator(f, 8, infinite),
                                pairs.getImposterGenerator(f, f, 8, infinite), 10)
        assert g1.indexable and g2.indexable
        np.random.seed(3)
        a = []
        try:
            for _ in range(60):
                a.append(next(g1))
        except StopIteration:
            assert not infinite
        sa = np.random.get_state()[1][:4].tolist()
        np.random.seed(3)
        t = g2.table()
        b = []
        try:
            for _ in range(60):
                li, ri, y = g2.next_indices()
                b.append(([t[li], t[ri]], y))
        except StopIteration:
            assert not infinite
        assert len(a) == len(b) and len(a) > 3 and sa == np.random.get_state()[1][:4].tolist()
        for (xa, ya), (xb, yb) in zip(a, b):
            assert np.array_equal(ya, yb) and np.array_equal(xa[0], xb[0]) and np.array_equal(xa[1], xb[1])
            assert (ya == 1).sum() == (ya == 0).sum() and len(ya) >= 10


def test_index_form_equals_the_round_by_round_form_over_foreign_sources():
    """the same sources wrapped so that the mix cannot look into them: the legacy row path, one round at a time"""
    f = _feats(25, 2)

    def foreign(g):
        while True:
            yield next(g)
    own = pairs.getGenerator(pairs.getNormalGenerator(f, 8), pairs.getNormalGenerator(f, 8), pairs.getImposterGenerator(f, f, 8), 12)
    leg = pairs.getGenerator(foreign(pairs.getNormalGenerator(f, 8)), foreign(pairs.getNormalGenerator(f, 8)),
                             foreign(pairs.getImposterGenerator(f, f, 8)), 12)
    assert own.indexable and not leg.indexable
    np.random.seed(11)
    a = [next(own) for _ in range(25)]
    np.random.seed(11)
    b = [next(leg) for _ in range(25)]
    for (xa, ya), (xb, yb) in zip(a, b):
        assert np.array_equal(ya, yb) and np.array_equal(xa[0], xb[0]) and np.array_equal(xa[1], xb[1])


def test_sources_moved_behind_the_mix_invalidate_its_schedule():
    f = _feats(20, 3)
    srcs = [pairs.getNormalGenerator(f, 8), pairs.getNormalGenerator(f, 8), pairs.getImposterGenerator(f, f, 8)]
    g = pairs.getGenerator(srcs[0], srcs[1], srcs[2], 10)
    ref_srcs = [pairs.getNormalGenerator(f, 8), pairs.getNormalGenerator(f, 8), pairs.getImposterGenerator(f, f, 8)]

    def foreign(s):
        while True:
            yield next(s)
    ref = pairs.getGenerator(*[foreign(s) for s in ref_srcs], 10)
    np.random.seed(1)
    next(g)
    next(srcs[0]), next(srcs[2])                              # someone else draws from two of the sources
    x1 = next(g)
    np.random.seed(1)
    next(ref)
    next(ref_srcs[0]), next(ref_srcs[2])
    x2 = next(ref)
    assert np.array_equal(x1[1], x2[1]) and np.array_equal(x1[0][0], x2[0][0]) and np.array_equal(x1[0][1], x2[0][1])

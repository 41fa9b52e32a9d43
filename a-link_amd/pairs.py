"""pairs — the pair-enumeration workload of an A-LINK iteration, as INDEX lists.

The reference materialises every pair's pixels on the host (reference code/readDFW.py:143-244,
code/readMTP.py:80-135): P pairs cost 2·P image copies although only a few dozen images are distinct
(SURVEY.md Appendix B).  Here a pair list is three integer arrays — `li`, `ri` (rows of one flat array
holding every distinct image / feature once) and `y` — built by array arithmetic; images are embedded
once and pairs are gathered by index on the device (alink_head_forward's index mode).

    PairTable                     people -> flat rows + per-person row ranges
    block_pairs / *_pair_indices  the enumeration orders of createMiniBatch / getNormalGenerator /
                                  getImposterGenerator as (li, ri, y)
    index_batches                 a pair list cut into generator batches (short tail dropped per sweep)
    balance_rows, Pending         the 1:1 class balancing and the accumulate-until-batch_size logic of
                                  readDFW.getGenerator / readMTP.getGenerator

The reference-named callables at the bottom (createMiniBatch, getNormalGenerator, getImposterGenerator,
getGenerator, getGeneratorMTP, splitDisguiseData) keep the reference's signatures and yield
`[left, right], Y` by gathering rows — they are what tests/golden/generators.npz and minibatch.npz,
recorded from the reference's own functions, are compared with.
"""
import numpy as np


class PairTable:
    """`people`: sequence of per-person arrays (k_i, ...).  `rows` is their concatenation,
    `start[i]:start[i+1]` person i's rows, `owner[r]` the person of row r."""

    def __init__(self, people, base=0):
        counts = np.fromiter((len(p) for p in people), dtype=np.int64, count=len(people))
        self.start = base + np.concatenate(([0], np.cumsum(counts)))
        self.counts = counts
        self.people = people

    @property
    def n_rows(self):
        return int(self.counts.sum())

    def rows(self):
        parts = [np.asarray(p) for p in self.people if len(p)]
        return np.concatenate(parts, axis=0) if parts else np.zeros((0,))

    def person_rows(self, i):
        return np.arange(self.start[i], self.start[i + 1], dtype=np.int64)


def block_pairs(ta, tb, label):
    """Blocks in (person a, person b) order; inside a block every row of a against every row of b,
    a-major.  `label(i, j)` gives the block's y.  This is the loop nest shared by createMiniBatch
    (code/readDFW.py:224-243) and getNormalGenerator (code/readDFW.py:146-152)."""
    li, ri, y = [], [], []
    for i in range(len(ta.counts)):
        a = ta.person_rows(i)
        if not len(a):
            continue
        for j in range(len(tb.counts)):
            b = tb.person_rows(j)
            if not len(b):
                continue
            li.append(np.repeat(a, len(b)))
            ri.append(np.tile(b, len(a)))
            y.append(np.full(len(a) * len(b), label(i, j), dtype=np.int64))
    if not li:
        z = np.zeros(0, np.int64)
        return z, z.copy(), z.copy()
    return np.concatenate(li), np.concatenate(ri), np.concatenate(y)


def _same(i, j):
    return 1 if i == j else 0


def createMiniBatchIndices(n_plain, n_dig):
    """n_plain[i], n_dig[i]: image counts per person.  (li, ri, y) into `concat(plain_0.., dig_0..)`
    in createMiniBatch's order: plain x disguised, then disguised x disguised; y = [i == j]."""
    tp = PairTable([range(int(n)) for n in n_plain])
    td = PairTable([range(int(n)) for n in n_dig], base=tp.n_rows)
    parts = [block_pairs(tp, td, _same), block_pairs(td, td, _same)]
    li, ri, y = (np.concatenate([p[k] for p in parts]) for k in range(3))
    return li.astype(np.int32), ri.astype(np.int32), y.reshape(-1, 1)


def normal_pair_indices(counts):
    """All (person i x person j) pairs of one people list, y = [i == j] (getNormalGenerator's sweep)."""
    t = PairTable([range(int(n)) for n in counts])
    return block_pairs(t, t, _same)


def imposter_pair_indices(plain_counts, imposter_counts):
    """Every plain row against every impostor row, label 0 (getImposterGenerator's sweep,
    code/readDFW.py:166-171): the person structure does not matter, so this is one outer product.
    Right indices are offset by the number of plain rows."""
    n_a, n_b = int(np.sum(plain_counts)), int(np.sum(imposter_counts))
    li = np.repeat(np.arange(n_a, dtype=np.int64), n_b)
    ri = n_a + np.tile(np.arange(n_b, dtype=np.int64), n_a)
    return li, ri, np.zeros(n_a * n_b, np.int64)


class index_batches(object):
    """Iterator of (li, ri, y[:, None]) slices of batch_size over a pair list; the remainder of a sweep that does not
    fill a batch is discarded before the next sweep starts (the reference resets its lists there).  Beyond a generator it
    knows, without drawing them, which of its coming batches hold a positive / a negative label (`flags_ahead`) and can
    `skip` batches: what mix_balanced uses to pass over the rounds the reference's getGenerator throws away
    (code/readDFW.py:191-193: `if minSamp == 0: continue` — with hundreds of persons ~400 draws per batch it keeps)."""

    def __init__(self, li, ri, y, batch_size, infinite=True):
        self.li, self.ri, self.y = li, ri, np.asarray(y)
        self.bs = int(batch_size)
        self.nb = len(self.y) // self.bs                       # full batches per sweep
        self.infinite = bool(infinite)
        self.k = 0                                             # batches drawn so far
        yb = self.y[:self.nb * self.bs].reshape(self.nb, self.bs) if self.nb else np.zeros((0, self.bs), self.y.dtype)
        self.has_pos = (yb == 1).any(axis=1)
        self.has_neg = (yb == 0).any(axis=1)
        self.pos_at = np.flatnonzero(self.has_pos)              # batches of a sweep that hold a positive, ascending

    def __iter__(self):
        return self

    def remaining(self):
        """batches left before StopIteration (None: never)"""
        return None if self.infinite else max(self.nb - self.k, 0)

    def __next__(self):
        if self.nb == 0:
            if self.infinite:
                raise RuntimeError("a pair list shorter than one batch never yields (the reference's generator would spin for ever)")
            raise StopIteration
        if not self.infinite and self.k >= self.nb:
            raise StopIteration
        b = self.k % self.nb
        self.k += 1
        s0 = b * self.bs
        return self.li[s0:s0 + self.bs], self.ri[s0:s0 + self.bs], self.y[s0:s0 + self.bs].reshape(-1, 1)

    def to_next_positive(self, ahead):
        """batches from (the batch `ahead` draws from now) to the first one that holds a positive; None: no batch ever does"""
        if len(self.pos_at) == 0:
            return None
        b = (self.k + ahead) % self.nb
        i = int(np.searchsorted(self.pos_at, b))
        return int(self.pos_at[i] - b) if i < len(self.pos_at) else int(self.pos_at[0] + self.nb - b)

    def holds_negative(self, ahead):
        return bool(self.has_neg[(self.k + ahead) % self.nb])

    def skip(self, n):
        self.k += int(n)


def balance_rows(y):
    """Row selection giving as many positives as negatives: min(#pos, #neg) of each, drawn without
    replacement — positives first, then negatives, two np.random.choice calls in that order
    (code/readDFW.py:189-199).  None when a class is absent."""
    flat = np.asarray(y).reshape(len(y), -1)[:, 0]
    groups = [np.flatnonzero(flat == 1), np.flatnonzero(flat == 0)]
    m = min(len(g) for g in groups)
    if m == 0:
        return None
    # np.random.choice(g, m, replace=False) IS g[np.random.permutation(len(g))[:m]] (numpy/random/mtrand.pyx, the legacy
    # RandomState.choice without p) minus its argument checks: the same draws (tests/test_pairs_generators.py)
    return np.concatenate([g[np.random.permutation(len(g))[:m]] for g in groups])


class Pending:
    """Rows waiting for a generator batch: append until at least batch_size, then flush everything."""

    def __init__(self):
        self.parts = []
        self.n = 0

    def add(self, left, right, y):
        self.parts.append((left, right, y))
        self.n += len(y)

    def flush(self):
        out = tuple(_cat([p[k] for p in self.parts]) for k in range(3))
        self.parts, self.n = [], 0
        return out


def _cat(parts):
    if hasattr(parts[0], "detach"):                     # device tensors stay on the device
        import torch
        return torch.cat(list(parts), dim=0)
    return np.concatenate(parts, axis=0)


def _take(rows, idx):
    if hasattr(rows, "detach"):
        import torch
        return rows[torch.as_tensor(np.asarray(idx), device=rows.device, dtype=torch.long)]
    return rows[idx]


def _rounds_to_skip(sources):
    """How many coming rounds of mix_balanced draw a joined batch WITHOUT a positive or without a negative label — rounds the
    reference draws, joins and throws away (`continue`, no random number used) — when every source can tell (`_Gathering` over
    index_batches).  None: a source cannot tell (a foreign generator) — draw round by round.  Stops at the first useful round or
    at a finite source's end.  A binary search per source and candidate round (the positives of an all-pairs sweep are 1 batch
    in ~P)."""
    if not all(isinstance(g, _Gathering) for g in sources):
        return None
    lab = list(sources)
    if len(lab) == 3:
        lab[2] = lab[1]                                       # the reference's (Y1, Y2, Y2)
    limit = None                                              # rounds until a finite source ends
    for g in sources:
        if g.index.nb == 0:
            return 0                                          # let next() raise what it raises
        r = g.index.remaining()
        if r is not None:
            limit = r if limit is None else min(limit, r)
    t = 0
    while True:
        steps = [g.index.to_next_positive(t) for g in lab]
        steps = [v for v in steps if v is not None]
        if not steps:                                         # no source ever yields a positive: every round is discarded
            if limit is None:
                raise RuntimeError("the balanced generator's sources hold no positive pair: the reference's loop would spin for ever")
            return limit
        t += min(steps)
        if limit is not None and t >= limit:
            return limit
        if any(g.index.holds_negative(t) for g in lab):
            return t
        t += 1                                                # a round of positives only: discarded too


class BalancedMix(object):
    """`sources`: generators of ([left, right], Y).  Per round: draw one batch from every source (stop
    when one is exhausted), join, balance classes, optionally transform the two sides, accumulate until
    batch_size rows are waiting.  With three sources the joined labels are (Y1, Y2, Y2): the reference
    repeats the second label block for the third source (code/readDFW.py:185) — kept.
    Rounds whose joined labels lack a class are dropped by the reference AFTER drawing and joining them, without touching
    the random stream; with this module's own sources they are passed over by index arithmetic instead (_rounds_to_skip):
    the same batches, the same np.random.choice calls, ~100x less host time at DFW scale (tools/custom_train_time.py).

    An iterator like the generator function it replaces (`next(gen)`, `gen.next()`, `for batch in gen`).  When every source
    is one of this module's own (rows of a table gathered by index) and nothing transforms the sides, the mix is INDEXABLE:
    `next_indices()` yields the coming batch as (li, ri, y) — rows of `table()`, the sources' tables one after another —
    without touching a feature; `next()` is then that gather.  SiameseNetwork.customTrainModel keeps the table on the
    device and ships only the indices (siamese.py)."""

    def __init__(self, sources, batch_size, transform=None):
        self.sources, self.batch_size, self.transform = list(sources), int(batch_size), transform
        self.waiting = Pending()
        own = all(isinstance(g, _Gathering) for g in self.sources)
        self.indexable = bool(own and transform is None and len(self.sources) > 0 and
                              len(set((tuple(g.rows.shape[1:]), str(g.rows.dtype), type(g.rows)) for g in self.sources)) == 1)
        if self.indexable:
            self._bases = np.concatenate(([0], np.cumsum([len(g.rows) for g in self.sources])))[:-1].astype(np.int64)
        self._table = None
        self._sched = None

    def __iter__(self):
        return self

    def table(self):
        """every source's rows, one table after another (what next_indices() points into)"""
        if self._table is None:
            self._table = _cat([g.rows for g in self.sources])
        return self._table

    def _skip_ahead(self):
        k = _rounds_to_skip(self.sources)
        while k:                                               # k is None for foreign generators
            for g in self.sources:
                g.index.skip(k)
            k = _rounds_to_skip(self.sources)

    # -- the index form: the rounds worth drawing are found in bulk, a window of rounds at a time -------------------------------
    def _useful_rounds(self):
        """Offsets (in rounds from now) of the coming rounds whose joined labels hold both classes, ascending — what
        _rounds_to_skip finds one round at a time, for a whole window by array arithmetic (the sources advance in lock-step:
        round t draws batch (k_i + t) mod nb_i of source i).  Cached until the sources move by other means."""
        ks = tuple(g.index.k for g in self.sources)
        c = self._sched
        if c is not None and c[0] == tuple(k - c[3] for k in ks) and c[3] < c[2]:
            return c
        idx = [g.index for g in self.sources]
        lab = [idx[0], idx[1]] if len(idx) == 3 else list(idx)        # the reference's (Y1, Y2, Y2)
        if any(ix.nb == 0 for ix in idx):
            return None
        limit = None
        for ix in idx:
            r = ix.remaining()
            if r is not None:
                limit = r if limit is None else min(limit, r)
        T = 4096
        while True:
            span = T if limit is None else min(T, limit)
            t = np.arange(span, dtype=np.int64)
            pos = np.zeros(span, bool)
            neg = np.zeros(span, bool)
            for ix in lab:
                b = (ix.k + t) % ix.nb
                pos |= ix.has_pos[b]
                neg |= ix.has_neg[b]
            useful = np.flatnonzero(pos & neg)
            if len(useful) or (limit is not None and span >= limit):
                break
            if not any(len(ix.pos_at) for ix in lab):
                raise RuntimeError("the balanced generator's sources hold no positive pair: the reference's loop would spin for ever")
            if T >= (1 << 26):
                raise RuntimeError("no round of the balanced generator holds both classes in %d rounds" % T)
            T *= 4
        # [source positions the offsets count from, useful offsets, rounds scanned, rounds consumed, next entry, end of a finite source]
        self._sched = [ks, useful, span, 0, 0, limit is not None and span >= limit]
        return self._sched

    def next_indices(self):
        """the next batch as (li, ri, y): int64 rows of table(), y of shape (n, 1).  StopIteration when a finite source ends."""
        if not self.indexable:
            raise TypeError("this mix is not indexable (a foreign source generator, or a transform of the sides)")
        w = self.waiting
        srcs = self.sources
        while True:
            c = self._useful_rounds()
            if c is None:
                for g in srcs:
                    next(g.index)                               # a source without a full batch: raises what it raises
                raise StopIteration
            ks, useful, span, done, j, ends = c
            if j >= len(useful):                                # nothing useful left in the window: pass over its rest
                for g in srcs:
                    g.index.skip(span - done)
                c[3] = span
                if ends:
                    for g in srcs:
                        next(g.index)                           # the exhausted source raises StopIteration
                continue
            t = int(useful[j])
            if t > done:
                for g in srcs:
                    g.index.skip(t - done)
            c[3], c[4] = t + 1, j + 1
            drawn = [next(g.index) for g in srcs]
            if len(drawn) == 3:
                y = np.concatenate((drawn[0][2], drawn[1][2], drawn[1][2]), axis=0)
            else:
                y = np.concatenate([d[2] for d in drawn], axis=0)
            keep = balance_rows(y)
            if keep is None:
                continue
            li = np.concatenate([d[0] + b for d, b in zip(drawn, self._bases)])
            ri = np.concatenate([d[1] + b for d, b in zip(drawn, self._bases)])
            w.add(li[keep], ri[keep], y[keep])
            if w.n >= self.batch_size:
                return w.flush()

    def __next__(self):
        if self.indexable:
            li, ri, yy = self.next_indices()
            t = self.table()
            return ([_take(t, li), _take(t, ri)], yy)
        waiting = self.waiting
        while True:
            self._skip_ahead()
            drawn = [next(g) for g in self.sources]
            labels = [d[1] for d in drawn]
            if len(labels) == 3:
                labels[2] = labels[1]
            y = np.concatenate(labels, axis=0)
            sides = [_cat([d[0][s] for d in drawn]) for s in (0, 1)]
            keep = balance_rows(y)
            if keep is None:
                continue
            sides = [_take(s, keep) for s in sides]
            if self.transform is not None:
                sides = self.transform(sides)
            waiting.add(sides[0], sides[1], y[keep])
            if waiting.n >= self.batch_size:
                left, right, yy = waiting.flush()
                return ([left, right], yy)

    next = __next__                                            # the reference calls gen.next() (Python 2)


mix_balanced = BalancedMix


# ---- the reference's names and signatures ---------------------------------------------------------------
class _Gathering(object):
    """generator of ([rows[li], rows[ri]], y) over an index_batches iterator (kept reachable: mix_balanced skips on it)"""

    def __init__(self, table, index):
        self.rows, self.index = table, index

    def __iter__(self):
        return self

    def __next__(self):
        li, ri, y = next(self.index)
        return [self.rows[li], self.rows[ri]], y

    next = __next__                                            # the reference calls gen.next() (Python 2)


def _gathering(table, idx_gen):
    return _Gathering(table, idx_gen)


def createMiniBatch(X_plain, X_dig):
    """reference code/readDFW.py:222-244 (= code/readMTP.py:123-135 with one list)."""
    tp, td = PairTable(X_plain), PairTable(X_dig, base=PairTable(X_plain).n_rows)
    rows = np.concatenate([tp.rows(), td.rows()], axis=0)
    parts = [block_pairs(tp, td, _same), block_pairs(td, td, _same)]
    li, ri, y = (np.concatenate([p[k] for p in parts]) for k in range(3))
    return [rows[li], rows[ri]], y.reshape(-1, 1)


def splitDisguiseData(X_dig, pre_ratio=0.5):
    """reference code/readDFW.py:212-219: each person's first int(k*pre_ratio) images / the rest."""
    cuts = [int(x.shape[0] * pre_ratio) for x in X_dig]
    return ([x[:c] for x, c in zip(X_dig, cuts)], [x[c:] for x, c in zip(X_dig, cuts)])


def getNormalGenerator(X_data, batch_size, infinite=True):
    """reference code/readDFW.py:143-160."""
    t = PairTable(X_data)
    li, ri, y = block_pairs(t, t, _same)
    return _gathering(t.rows(), index_batches(li, ri, y, batch_size, infinite))


def getImposterGenerator(X_plain, X_imposter, batch_size, infinite=True):
    """reference code/readDFW.py:163-177."""
    ta, tb = PairTable(X_plain), PairTable(X_imposter)
    li, ri, y = imposter_pair_indices(ta.counts, tb.counts)
    rows = np.concatenate([ta.rows(), tb.rows()], axis=0)
    return _gathering(rows, index_batches(li, ri, y, batch_size, infinite))


def getGenerator(norGen, normImpGen, impGen, batch_size, type=0, val_ratio=0.2):
    """reference code/readDFW.py:180-209.  Ends (StopIteration) when a finite source generator is
    exhausted, where the Python-3 copy yields None (code/readDFW3.py)."""
    return mix_balanced([norGen, normImpGen, impGen], batch_size)


def getGeneratorMTP(datGen, batch_size, resize_res=None, featurize=None):
    """reference code/readMTP.py:80-113: balance, optionally resize (bilinear, device kernel) and
    featurize each source batch, accumulate to batch_size.
    Over one of this module's own sources (rows of a table gathered by index) the resize / featurize run ONCE, on the
    table: both are per-image maps (an image resizes — and embeds — to the same values whatever batch it arrives in), so
    gathering rows of the transformed table IS transforming the gathered rows; the reference resizes every copy of every
    image again in every round (a pair batch holds each image many times).  The mix is then indexable like the DFW one."""
    def transform(sides):
        if resize_res:
            from . import noise as _noise
            sides = [np.asarray(_noise.resize_images(s, resize_res)) for s in sides]
        if featurize:
            sides = [featurize.process(s) for s in sides]
        return sides
    if (resize_res or featurize) and isinstance(datGen, _Gathering) and len(datGen.rows):
        rows = datGen.rows
        step = 4096                                   # the table goes through the device in chunks (pool-sized tables)
        parts = [transform([rows[i:i + step]])[0] for i in range(0, len(rows), step)]
        return mix_balanced([_Gathering(_cat([np.asarray(p) if not hasattr(p, "detach") else p for p in parts]), datGen.index)], batch_size)
    return mix_balanced([datGen], batch_size, transform if (resize_res or featurize) else None)

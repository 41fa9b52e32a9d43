"""CPU, gloo, world sizes 2 and 3: the multi-rank A-LINK iteration (alink_loop.alink_iteration / run_alink_dfw with `group`,
BASELINE configs[3] / configs[4]; reference loop code/ALINK_arc.py:142-254, fine-tune code/siamese.py:52-58).

The loop is duck-typed over its models, so the control path that shards an iteration — row ranges per rank, the
prediction all-gather, replicated selection, settle requests served by the owner of a row, the fine-tune rows gathered from
their owners, rank 0's host randomness on every rank, the fine-tune through KerasFitMixin.fit(dp_group) — runs here
with NumPy stand-ins for the device models: a feature model whose rows do not depend on the batch they arrive in (as the
HIP backbone's do not), a pair head on the oracle's arithmetic under the PRODUCT's fit(), noise keyed by the global row.
Every rank must end with the query lists, counts, fine-tune sets and student weights of the single-process loop, bit for
bit.  (The device kernels' own row-range invariance and the same comparison on the GPU: tests/test_gpu_noise.py,
tests/test_gpu_distributed.py.)"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import a_link_amd  # noqa: F401
from a_link_amd import alink_loop as AL
from a_link_amd import committee, pairs, settle, siamese
from a_link_amd.head import KerasFitMixin

SIZE = (8, 8)
D = 24


class FakeFeature(object):
    """(n, H, W, 3) -> (n, D) unit rows; a row's value does not depend on the batch (sums over a middle axis add slices
    in order).  process_screen: the same map on pixels rounded to float16 multiples of 1/4 — a coarser, batch-invariant
    arithmetic, like the 16-bit screening mode."""

    def __init__(self, seed=3):
        self.W = np.random.RandomState(seed).randn(SIZE[0] * SIZE[1] * 3, D).astype(np.float64) / 64.0
        self.calls = {"exact": 0, "screen": 0}

    def _embed(self, X):
        X = np.asarray(X, np.float64).reshape(len(X), -1)
        e = (X[:, :, None] * self.W[None]).sum(axis=1)
        return (e / np.sqrt((e * e).sum(axis=1, keepdims=True))).astype(np.float32)

    def process(self, X):
        self.calls["exact"] += len(X)
        if len(X) == 0:
            return np.zeros((0, D), np.float32)
        return self._embed(np.asarray(X, np.float32))

    def process_screen(self, X):
        self.calls["screen"] += len(X)
        if len(X) == 0:
            return np.zeros((0, D), np.float32)
        return self._embed(np.round(np.asarray(X, np.float32) * 0.5) * 2.0)


class FakeNoise(object):
    """x + N(0, sigma) with every row's draw keyed by (stream seed, call, GLOBAL row): the contract of noise.py's classes"""
    supports_rows = True

    def __init__(self, seed, sigma):
        self._seed, self._calls, self.sigma = int(seed), 0, float(sigma)

    def stream_state(self):
        return (self._seed, self._calls)

    def set_stream_state(self, st):
        self._seed, self._calls = int(st[0]), int(st[1])

    def addPairNoise(self, image_pairs, target_labels, rows=None):
        first = 0 if rows is None else int(rows[0])
        out = []
        for side in image_pairs:
            call = self._calls
            self._calls += 1
            x = np.array(side, np.float32, copy=True)
            for i in range(len(x)):
                r = np.random.RandomState([self._seed & 0x7FFFFFFF, call, first + i])
                x[i] += (self.sigma * r.randn(*x[i].shape)).astype(np.float32)
            out.append(x)
        return out


class CpuBagging(committee.Bagging):
    def resize(self, images, new_size):          # the test's images already have the target size
        return np.asarray(images)


class CpuNet(KerasFitMixin):
    """the oracle's head arithmetic under the PRODUCT's fit(): Keras control flow, dp_group, rank 0's shuffles"""

    def __init__(self, seed, scale=1.0):
        from oracle import siamese_head as O
        self.m = O.HeadModel(D, 16, 8, lr=0.1, seed=seed)
        ws = self.m.get_weights()
        ws[4] = ws[4] * np.float32(scale)          # spread the probabilities of a fresh head over (0, 1)
        self.m.set_weights(ws)
        self.steps = 0

    def get_weights(self):
        return self.m.get_weights()

    def get_lr(self):
        return self.m.get_lr()

    def set_lr(self, lr):
        self.m.set_lr(lr)

    def train_on_batch(self, x, y, class_weight=None, sample_weight=None):
        self.steps += 1
        return self.m.train_on_batch(x, y, class_weight=class_weight, sample_weight=sample_weight)

    def test_on_batch(self, x, y):
        return self.m.test_on_batch(x, y)

    def predict(self, X, batch_size=1024, verbose=0):
        # row-invariant forward (a row's probability must not depend on which rows share its batch)
        W1, b1, W2, b2, W3, b3 = [w.astype(np.float64) for w in self.m.get_weights()]
        h = np.abs(np.asarray(X[0], np.float64) - np.asarray(X[1], np.float64))
        h = np.maximum((h[:, :, None] * W1[None]).sum(axis=1) + b1, 0.0)
        h = np.maximum((h[:, :, None] * W2[None]).sum(axis=1) + b2, 0.0)
        z = (h[:, :, None] * W3[None]).sum(axis=1) + b3
        z = np.exp(z - z.max(axis=1, keepdims=True))
        return (z / z.sum(axis=1, keepdims=True)).astype(np.float32)


class CpuStudent(siamese.SiameseNetwork):
    """siamese.SiameseNetwork (its finetune / predict / preprocess are the product's) on a CPU net"""

    def __init__(self, seed, scale=1.0):
        self.siamese_net = CpuNet(seed, scale)
        self.modelName, self.shape, self.learningRate = "cpu", (D,), 0.1

    def save(self, customName=None):
        pass


def _people(n, seed, lo=2, hi=3):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 256, (rng.randint(lo, hi + 1),) + SIZE + (3,)).astype(np.float32) for _ in range(n)]


def _run(group, rank, screen, tiny=False):
    """one full loop; returns what must agree between ranks and with the single-process run.  tiny: one person with one plain and
    one disguised image per iteration — P = 2 pair rows, fewer than a world of 3 has ranks (an empty shard every iteration)"""
    flags = AL.Flags(alink_bs=1 if tiny else 3, batch_send=2 if tiny else 6, disparity_ratio=1.0 if tiny else 0.6, eps=0.0005, ft_epochs=2,
                     mixture_ratio=2, out_model="", screen_settle=screen)
    X_plain, X_dig = (_people(5, 1, 1, 1), _people(5, 2, 1, 1)) if tiny else (_people(9, 1), _people(9, 2))
    conv = FakeFeature()
    if not screen:
        conv.process_screen = None
    student = CpuStudent(7, scale=60.0)
    ens = [CpuStudent(100 + i, scale=60.0) for i in range(2)]
    # rank 0 carries the seeds of the single-process run; every other rank starts from different host randomness and
    # different noise streams — the iteration must make rank 0's theirs
    nz = [FakeNoise(1000 + i + 77 * rank, s) for i, s in enumerate((6.0, 14.0))]
    bag = CpuBagging(ens, nz)
    feats_plain = [conv.process(p) for p in X_plain]
    gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                             pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
    np.random.seed(5 + 1000 * rank)
    conv.calls = {"exact": 0, "screen": 0}
    sets = []
    o1, o2 = AL.selection.select_queries, settle.select_queries_settled
    AL.selection.select_queries = lambda *a, **k: (lambda r: (sets.append(list(r[0])), r)[1])(o1(*a, **k))
    settle.select_queries_settled = lambda *a, **k: (lambda r: (sets.append(list(r[0])), r)[1])(o2(*a, **k))
    try:
        st = AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, SIZE, col=0, verbose=0, on_device=False,
                              group=group)
    finally:
        AL.selection.select_queries, settle.select_queries_settled = o1, o2
    return {"active": st.active_count, "un": st.un_size, "finetunes": st.finetunes, "sets": sets,
            "weights": student.siamese_net.get_weights(), "steps": student.siamese_net.steps, "calls": dict(conv.calls),
            "settle_info": st.settle_info, "hist": [h["loss"] for h in st.history]}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for screen in (False, True):
            out[screen] = _run(dist.group.WORLD, rank, screen)
        out["tiny"] = _run(dist.group.WORLD, rank, True, tiny=True)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multirank_loop_equals_single_process_loop(world):
    want = {screen: _run(None, 0, screen) for screen in (False, True)}
    want_tiny = _run(None, 0, True, tiny=True)
    assert want[False]["finetunes"] >= 1 and len(want[False]["sets"]) >= 3 and sum(len(s) for s in want[False]["sets"]) >= 10, \
        "test data must select queries and trigger a fine-tune"
    assert want[True]["sets"] == want[False]["sets"]            # screen-then-settle reaches the all-exact selection here too
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):                       # P = 2 rows per iteration: with three ranks one shard is always empty
        g = res[r]["tiny"]
        assert (g["active"], g["un"], g["finetunes"], g["sets"]) == (want_tiny["active"], want_tiny["un"], want_tiny["finetunes"], want_tiny["sets"]), r
        assert all(np.array_equal(a, b) for a, b in zip(g["weights"], want_tiny["weights"])), r
    assert want_tiny["un"] == 10
    for screen in (False, True):
        w = want[screen]
        for r in range(world):
            g = res[r][screen]
            assert (g["active"], g["un"], g["finetunes"], g["steps"]) == (w["active"], w["un"], w["finetunes"], w["steps"]), (screen, r)
            assert g["sets"] == w["sets"], (screen, r)
            assert g["hist"] == w["hist"], (screen, r)
            for a, b in zip(g["weights"], w["weights"]):
                assert np.array_equal(a, b), (screen, r)
        # the noisy embeddings — the bulk of an iteration — are SHARED: together the ranks convert what one process converts
        # (+ the replicated clean pass), not world times as much
        key = "screen" if screen else "exact"
        total = sum(res[r][screen]["calls"][key] for r in range(world))
        clean = w["calls"]["exact"] - (0 if screen else 0)
        if screen:
            assert total == w["calls"]["screen"], (total, w["calls"])
            per_rank = [res[r][screen]["calls"]["screen"] for r in range(world)]
        else:
            # exact calls = replicated clean pass (unique images, once per rank) + the sharded noisy rows
            n_unique = sum(len(p) for p in _people(9, 1)) + sum(len(p) for p in _people(9, 2))
            assert total == w["calls"]["exact"] + (world - 1) * n_unique, (total, w["calls"], n_unique)
            per_rank = [res[r][screen]["calls"]["exact"] - n_unique for r in range(world)]
        assert max(per_rank) - min(per_rank) <= 2 * 2 * len(w["sets"]) * 1, per_rank       # contiguous shards differ by <= 1 row per noise, side and iteration
        if screen:
            info = res[0][screen]["settle_info"]
            assert info and info[0]["world"] == world and "rows_of_this_rank" in info[0]


def test_row_shards_subsets_and_gathers_world_3():
    """RowShards on its own: uneven shards, an empty request, a rank that owns nothing of a request"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shards_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _shards_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from a_link_amd import distributed as Dm
        ok = True
        for P in (10, 2, 0):
            sh = Dm.RowShards(P, dist.group.WORLD)
            table = np.arange(P * 3, dtype=np.float32).reshape(P, 3) * 1.5
            ok &= np.array_equal(sh.all_rows(table[sh.lo:sh.hi]), table)
            reqs = [np.array([0, 3, 4, 9]), np.array([], np.int64), np.array([8, 9]), np.arange(P)]
            reqs = [r[r < P] for r in reqs]
            got = sh.subsets(reqs, [table[sh.lo:sh.hi][sh.owned(r)] for r in reqs], (3,))
            ok &= all(np.array_equal(g, table[r]) for g, r in zip(got, reqs))
            ok &= sh.bcast({"x": rank}) == {"x": 0}
            ok &= sh.all_true(True) and not sh.all_true(rank != 1)
            ok &= sh.same_everywhere("a") and not sh.same_everywhere(rank)
        # a rank that could not produce its part says so IN the exchange: every rank raises, nobody waits
        sh = Dm.RowShards(9, dist.group.WORLD)
        if rank == 2:
            sh.fail("boom")
        try:
            sh.all_rows(np.zeros((sh.hi - sh.lo, 2), np.float32))
            ok = False
        except RuntimeError as e:
            ok &= "[2]" in str(e) and (("boom" in str(e)) == (rank == 2))
        ok &= np.array_equal(sh.all_rows(np.ones((sh.hi - sh.lo, 2), np.float32)), np.ones((9, 2), np.float32))      # and the next exchange is clean

        # merge_calibration: every rank ends with the elementwise minimum of the ranks' scale exponents (a scale only goes down)
        class FakeBackbone(object):
            def __init__(self, e):
                self.st = {"dtype": "f16x2", "scale_exponents": list(e)}

            def state(self):
                return dict(self.st)

            def load_state(self, st):
                self.st = dict(st)
        bb = FakeBackbone([5 - rank, 3 + rank, 7])
        changed = Dm.merge_calibration([bb], dist.group.WORLD)
        ok &= bb.state()["scale_exponents"] == [3, 3, 7] and changed == (rank != 1 or True) and sh.same_everywhere(bb.state())
        # sync_host_randomness: rank 0's NumPy stream and noise stream states on every rank
        nzs = [FakeNoise(100 + rank, 1.0)]
        nzs[0]._calls = rank
        np.random.seed(rank)
        AL.sync_host_randomness(nzs, sh)
        ok &= nzs[0].stream_state() == (100, 0) and sh.same_everywhere(float(np.random.rand()))
        # ranks whose feature models carry different calibration states must not start an iteration: every rank raises
        try:
            AL.alink_iteration(AL.LoopState(), AL.Flags(), None, np.zeros((4, 1)), None, None, [], None, None, None, None, SIZE,
                               group=dist.group.WORLD, calibration_of=lambda: {"scale_exponents": [rank]})
            ok = False
        except RuntimeError as e:
            ok &= "calibration state differs between ranks" in str(e)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


# ---- a rank that fails in the BULK phase, and a rank that re-calibrates itself ---------------------------------------
class _FailingNoise(FakeNoise):
    """raises for the rank whose rows include a chosen global row (a data-dependent failure: Poisson's lam < 0, say)"""

    def __init__(self, seed, sigma, bad_row):
        FakeNoise.__init__(self, seed, sigma)
        self.bad_row = bad_row

    def addPairNoise(self, image_pairs, target_labels, rows=None):
        first = 0 if rows is None else int(rows[0])
        if first <= self.bad_row < first + len(image_pairs[0]):
            raise ValueError("lam < 0 in row %d" % self.bad_row)
        return FakeNoise.addPairNoise(self, image_pairs, target_labels, rows=rows)


class _RecalibratingFeature(FakeFeature):
    """a feature model with a calibration state (one scale exponent) that rank `who` lowers by itself the first time it converts
    noisy rows — what an f16x2 backbone does when a batch leaves its range; `model.model` is what merge_calibration talks to"""

    def __init__(self, who, rank):
        FakeFeature.__init__(self)
        self.exp, self.who, self.rank, self.tripped, self.merged = [7], who, rank, False, 0
        outer = self

        class _BB(object):
            dtype = "f16x2"

            def state(self):
                return {"dtype": "f16x2", "scale_exponents": list(outer.exp)}

            def load_state(self, st):
                outer.exp = list(st["scale_exponents"])
                outer.merged += 1
        self.model = type("M", (), {"model": _BB()})()
        self.process_screen = None

    def process(self, X):
        if len(X) > 40 and self.rank == self.who and not self.tripped:       # the noisy rows (the clean pass is a few dozen images)
            self.tripped = True
            self.exp = [5]
        return FakeFeature.process(self, X)


def _advice_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        X_plain, X_dig = _people(6, 1), _people(6, 2)
        # (a) the noise of ONE rank raises: every rank must raise in the exchange that follows, none may wait
        for screen in (False, True):
            flags = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=2, mixture_ratio=2, out_model="",
                             screen_settle=screen)
            conv = FakeFeature()
            if not screen:
                conv.process_screen = None
            student = CpuStudent(7, scale=60.0)
            ens = [CpuStudent(100 + i, scale=60.0) for i in range(2)]
            nz = [FakeNoise(1000, 6.0), _FailingNoise(1001, 14.0, bad_row=1)]              # row 1 belongs to rank 0
            feats_plain = [conv.process(p) for p in X_plain]
            gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                                     pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
            try:
                AL.run_alink_dfw(flags, conv, CpuBagging(ens, nz), nz, student, X_plain, X_dig, gen, SIZE, col=0, verbose=0, on_device=False,
                                 group=dist.group.WORLD)
                out["fail_%s" % screen] = "no error"
            except RuntimeError as e:
                out["fail_%s" % screen] = str(e)
            dist.barrier()
        # (b) one rank re-calibrates during the first iteration: scales merged, the iteration repeated, the loop completes — and
        # every rank ends with the student the single-process loop (calibrated to the merged scales from the start) ends with
        flags = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=2, mixture_ratio=2, out_model="")
        conv = _RecalibratingFeature(who=1, rank=rank)
        student = CpuStudent(7, scale=60.0)
        ens = [CpuStudent(100 + i, scale=60.0) for i in range(2)]
        nz = [FakeNoise(1000 + i, s) for i, s in enumerate((6.0, 14.0))]
        feats_plain = [FakeFeature.process(conv, p) for p in X_plain]
        gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                                 pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
        np.random.seed(5)
        st = AL.run_alink_dfw(flags, conv, CpuBagging(ens, nz), nz, student, X_plain, X_dig, gen, SIZE, col=0, verbose=0, on_device=False,
                              group=dist.group.WORLD)
        out["recal"] = {"recalibrations": st.recalibrations, "exp": list(conv.exp), "iterations": st.iterations, "un": st.un_size,
                        "active": st.active_count, "finetunes": st.finetunes, "weights": student.siamese_net.get_weights()}
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_and_a_recalibrating_rank_do_not_strand_their_peers():
    """ADVICE r5.  (a) A data-dependent failure in the bulk phase of alink_iteration(group=) — here the noise of the rank that owns
    pair row 1 raises — is carried into the next exchange: BOTH ranks raise a RuntimeError naming rank 0 (only rank 0's message holds
    the cause), with and without screen-then-settle; nobody blocks in a collective.  (b) A rank whose feature model lowers its
    split-precision scales by itself during an iteration makes every rank roll the iteration back, take the elementwise minimum
    of the scales, wind the random streams back and run it again — once; the loop then finishes with the counters and the student
    of an undisturbed single-process loop."""
    # the single-process loop (b) must reproduce: no re-calibration happens (rank 0 is never `who`)
    X_plain, X_dig = _people(6, 1), _people(6, 2)
    flags = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=2, mixture_ratio=2, out_model="")
    conv = _RecalibratingFeature(who=1, rank=0)
    student = CpuStudent(7, scale=60.0)
    ens = [CpuStudent(100 + i, scale=60.0) for i in range(2)]
    nz = [FakeNoise(1000 + i, s) for i, s in enumerate((6.0, 14.0))]
    feats_plain = [FakeFeature.process(conv, p) for p in X_plain]
    gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                             pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
    np.random.seed(5)
    want = AL.run_alink_dfw(flags, conv, CpuBagging(ens, nz), nz, student, X_plain, X_dig, gen, SIZE, col=0, verbose=0, on_device=False)
    want_w = student.siamese_net.get_weights()
    assert want.finetunes >= 1 and want.recalibrations == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_advice_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for screen in (False, True):
        for r in range(2):
            msg = res[r]["fail_%s" % screen]
            assert "rank(s) [0] failed before this exchange" in msg, (screen, r, msg)
            assert ("lam < 0 in row 1" in msg) == (r == 0), (screen, r, msg)
    for r in range(2):
        got = res[r]["recal"]
        assert got["recalibrations"] == 1 and got["exp"] == [5], got
        assert (got["iterations"], got["un"], got["active"], got["finetunes"]) == (want.iterations, want.un_size, want.active_count, want.finetunes)
        for a, b in zip(got["weights"], want_w):
            assert np.array_equal(a, b)

"""CPU, world_size 2, gloo: the sharding / merge / gradient-normalisation logic of distributed.py.
(The device kernels themselves are covered by the -m gpu tests; here the per-rank compute is the
NumPy oracle so that the N > 1 control path is exercised without a GPU.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import a_link_amd  # noqa: F401
from a_link_amd import distributed as D


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 100000):
        for w in (1, 2, 3, 8):
            parts = [D.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [h - l for l, h in parts]
            assert max(sizes) - min(sizes) <= 1


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
        from oracle import siamese_head as O
        # ---- 1. pool top-k: shard scores, local top-k, merge == global top-k (ties -> lower index)
        rng = np.random.RandomState(0)
        scores = rng.rand(1000).astype(np.float32)
        scores[[10, 500, 900]] = 2.0                      # a three-way tie at the top, across shards
        k = 16
        lo, hi = D.shard_range(len(scores), rank, world)
        loc = scores[lo:hi]
        order = np.lexsort((np.arange(lo, hi), -loc))[:k]
        v, i = D.merge_topk(torch.from_numpy(loc[order]), torch.from_numpy(order + lo), k, largest=True)
        want = np.lexsort((np.arange(1000), -scores))[:k]
        ok1 = np.array_equal(i.numpy(), want) and np.array_equal(v.numpy(), scores[want])
        # smallest-k with a short shard (k > local size on purpose)
        v2, i2 = D.merge_topk(torch.from_numpy(np.sort(loc)[:5]), torch.from_numpy(np.argsort(loc, kind="stable")[:5] + lo),
                              8, largest=False)
        cand = np.concatenate([np.argsort(scores[l:h], kind="stable")[:5] + l for l, h in D.dp_batch_slices(1000, world)])
        want2 = cand[np.lexsort((cand, scores[cand]))][:8]
        ok2 = np.array_equal(i2.numpy(), want2)

        # ---- 2. DP gradient normalisation: per-rank grads with the GLOBAL denominator, summed,
        # equal the full-batch Keras gradients (class-weighted, one zero weight)
        ws = O.init_weights(64, 128, 32, seed=3)
        rs = np.random.RandomState(1)
        L, R = rs.randn(13, 64).astype(np.float32), rs.randn(13, 64).astype(np.float32)
        y = O.to_categorical(rs.randint(0, 2, 13))
        sw = np.where(y[:, 1] > 0, 2.0 / 3, 1.0 / 3).astype(np.float32)
        sw[4] = 0.0
        full, loss_full, _ = O.gradients(ws, L, R, y, sw)
        denom = float((sw != 0).sum())
        lo, hi = D.shard_range(13, rank, world)
        # local gradients of sum_i w_i l_i / denom  ==  oracle gradients with weights w_i * n_loc_nonzero / denom
        nloc = float((sw[lo:hi] != 0).sum())
        loc_g, loc_loss, _ = O.gradients(ws, L[lo:hi], R[lo:hi], y[lo:hi], sw[lo:hi])
        flat = torch.from_numpy(np.concatenate([g.ravel() for g in loc_g]) * np.float32(nloc / denom))
        D.allreduce_sum_(flat)
        ref = np.concatenate([g.ravel() for g in full])
        ok3 = np.allclose(flat.numpy(), ref, rtol=1e-5, atol=1e-7)
        lt = torch.tensor([loc_loss * nloc / denom], dtype=torch.float64)
        D.allreduce_sum_(lt)
        ok4 = abs(lt.item() - loss_full) < 1e-6

        # ---- 3. sharded embedding gather keeps order (fake feature model: row id -> embedding)
        class Fake(object):
            def process(self, X):
                return np.repeat(np.asarray(X, np.float32).reshape(len(X), -1)[:, :1], 4, axis=1)
        X = np.arange(11, dtype=np.float32).reshape(11, 1, 1, 1)
        E, (l0, h0) = D.embed_pool_sharded(Fake(), X)
        ok5 = np.array_equal(E.numpy()[:, 0], np.arange(11)) and (l0, h0) == D.shard_range(11, rank, world)
        # ---- 4. merge_topk refuses what it cannot carry or order: an index beyond int32, candidates not sorted
        # best-first, ties not in ascending index order (both ranks raise before the collective: no hang)
        ok6 = True
        for vals, idx in (([0.9, 0.5], [2 ** 31, 7]), ([0.5, 0.9], [1, 2]), ([0.5, 0.5], [9, 3])):
            try:
                D.merge_topk(torch.tensor(vals), torch.tensor(idx, dtype=torch.int64), 4, largest=True)
                ok6 = False
            except ValueError:
                pass
        # real candidates that score -inf tie with the padding and must still come back (3 real ones per rank, k = 8)
        v3, i3 = D.merge_topk(torch.tensor([1.0, float("-inf"), float("-inf")]), torch.tensor([0, 1, 2]) + 10 * rank, 8, largest=True)
        ok7 = i3.numpy().tolist() == [0, 10, 1, 2, 11, 12] and v3.numpy()[:2].tolist() == [1.0, 1.0]
        q.put((rank, ok1, ok2, ok3, ok4, ok5, ok6, ok7))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    for r in res:
        assert all(r[1:]), r


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks through torch.distributed.run as a child process
    (never an exec of a process that touched the GPU).  On this CPU-only box the ranks must fail with bench.py's own
    "needs a ROCm device" — not with the old WORLD_SIZE assertion — and the parent must report a non-zero code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher's failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180, env=env)
    assert p.returncode != 0
    assert "needs a ROCm device" in p.stderr and "WORLD_SIZE=" not in p.stderr
    assert p.stdout.strip() == ""


def _worker8(rank, world, port, q):
    """Eight ranks (the node the driver scales to): shards shorter than the world, several EMPTY shards, eight candidate
    lists of different lengths in one merge."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = []
        # ---- n < world: 5 pool images over 8 ranks -> ranks 5..7 hold nothing, and say so
        class Fake(object):
            def process(self, X):
                if len(X) == 0:                      # what ArcFace.process returns for an empty shard
                    return np.zeros((0, 4), np.float32)
                return np.repeat(np.asarray(X, np.float32).reshape(len(X), -1)[:, :1], 4, axis=1).reshape(len(X), 4)
        X = np.arange(5, dtype=np.float32).reshape(5, 1, 1, 1)
        E, (l0, h0) = D.embed_pool_sharded(Fake(), X)
        ok.append(np.array_equal(E.numpy()[:, 0], np.arange(5)) and (l0, h0) == D.shard_range(5, rank, world) and (h0 - l0 == 0) == (rank >= 5))
        # ---- top-k over 8 candidate lists: pool of 37 pairs (ranks hold 5,5,5,5,5,4,4,4), k = 16 > any local length,
        # ties across ranks; then a pool of 3 pairs (five empty ranks), k = 8 > the whole pool
        for n, k in ((37, 16), (3, 8), (1000, 64)):
            rng = np.random.RandomState(n)
            scores = rng.rand(n).astype(np.float32)
            scores[rng.randint(0, n, max(1, n // 4))] = 0.5
            lo, hi = D.shard_range(n, rank, world)
            loc = scores[lo:hi]
            for largest in (True, False):
                order = np.lexsort((np.arange(lo, hi), -loc if largest else loc))[:k]
                v, i = D.merge_topk(torch.from_numpy(loc[order]), torch.from_numpy(order + lo), k, largest=largest)
                want = np.lexsort((np.arange(n), -scores if largest else scores))[:k]
                ok.append(np.array_equal(i.numpy(), want) and np.array_equal(v.numpy(), scores[want]))
        # ---- batch slices cover a 16-row fine-tune batch at 8 ranks (2 rows each) and a 3-row batch (five empty slices)
        ok.append(D.dp_batch_slices(16, 8) == [(2 * r, 2 * r + 2) for r in range(8)])
        s3 = D.dp_batch_slices(3, 8)
        ok.append(s3[:3] == [(0, 1), (1, 2), (2, 3)] and all(a == b == 3 for a, b in s3[3:]))
        # ---- screen-then-settle across 8 ranks, three of them with an empty shard
        from a_link_amd import settle
        n, g, k = 5, 4, 7
        rs = np.random.default_rng(0)
        p = np.clip(0.5 + rs.normal(0, 0.1, n * g), 0.01, 0.99)
        ps = np.clip(p + rs.normal(0, 1e-3, n * g), 0, 1)
        lo, hi = D.shard_range(n, rank, world)
        pl, psl = p[lo * g:hi * g], ps[lo * g:hi * g]
        sc = lambda x: settle._score_of_u(np.abs(np.asarray(x, np.float64) - 0.5), "entropy").astype(np.float32)
        fn = lambda imgs: ((imgs[:, None] * g + np.arange(g)).ravel(), pl[(imgs[:, None] * g + np.arange(g)).ravel()],
                           sc(pl[(imgs[:, None] * g + np.arange(g)).ravel()]))
        vals, idx, info = settle.settle_topk(psl, sc(psl), np.repeat(np.arange(hi - lo), g), hi - lo, fn, k, comm=settle.make_comm(), base=lo * g)
        ok.append(np.array_equal(idx, np.lexsort((np.arange(n * g), -sc(p)))[:k]))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_world_size_8_gloo_short_and_empty_shards():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert len(res) == 8
    for r, ok in res:
        assert all(ok), (r, ok)


# ---- the data-parallel step's control flow over the ORACLE's SmallRes arithmetic (world 2 and 3, gloo) -------------------------
class _OracleSmallRes(object):
    """oracle.smallres under the product's distributed.dp_train_on_batch: the four methods a model gives it (grads_tensor,
    dp_begin, dp_local_grads, dp_apply) restated on the CPU — gradients of a slice with the global normaliser, dropout masks
    drawn for the WHOLE batch from the step's seed and cut to the slice's rows, one flat buffer [params | 4 metric floats]"""
    DP_SHARD_MIN_ROWS = 6

    def __init__(self, ws, lr=0.1):
        from oracle import smallres as OS
        self.OS = OS
        self.m = OS.SmallResModel(ws, lr=lr)
        self.sizes = [int(np.prod(w.shape)) for w in self.m.ws]
        self.flat = torch.zeros(sum(self.sizes) + 4, dtype=torch.float32)
        self.shapes = ((7, 7, 32), (2, 2, 64))                  # 16 x 16 inputs: after pool 1 / pool 2
        self.training_dropout = True

    def _masks(self, n, seed):
        e1, e2 = (int(np.prod(s)) for s in self.shapes)
        return (np.random.RandomState(seed).rand(2 * n * e1 + 2 * n * e2) >= 0.25).astype(np.uint8)

    def _rows(self, masks, n, lo, hi):
        e1, e2 = (int(np.prod(s)) for s in self.shapes)
        m1, m2 = masks[:2 * n * e1].reshape(2 * n, e1), masks[2 * n * e1:].reshape(2 * n, e2)
        return np.concatenate([m1[lo:hi].ravel(), m1[n + lo:n + hi].ravel(), m2[lo:hi].ravel(), m2[n + lo:n + hi].ravel()])

    def train_on_batch(self, x, y, class_weight=None, sample_weight=None):
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        n = len(y)
        out, _ = self.m.train_on_batch(x, y, sample_weight=sample_weight, masks=self._masks(n, seed), mask_shapes=self.shapes)
        return out

    def grads_tensor(self, with_metrics=False):
        return self.flat if with_metrics else self.flat[:-4]

    def dp_begin(self, n, group):
        import torch.distributed as dist
        box = [int(np.random.randint(0, 2 ** 31 - 1))]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]

    def dp_local_grads(self, x, y, w_all, lo, hi, n, grad_scale, m, seed):
        OS = self.OS
        t = [torch.tensor(w, requires_grad=True) for w in self.m.ws]
        p = OS.forward(t, x[0][lo:hi], x[1][lo:hi], self._rows(self._masks(n, seed), n, lo, hi), self.shapes)
        yt = torch.as_tensor(np.asarray(y[lo:hi], np.float32))
        pc = torch.clamp(p, 1e-7, 1 - 1e-7)
        z = torch.log(pc / (1 - pc))
        l = (torch.clamp(z, min=0) - z * yt + torch.log1p(torch.exp(-z.abs()))).mean(dim=1)
        w = torch.ones(hi - lo) if w_all is None else torch.as_tensor(w_all[lo:hi])
        loss = (l * w).sum() * grad_scale                       # this slice's share of the global weighted mean
        loss.backward()
        self.flat[:-4] = torch.cat([g.grad.reshape(-1) for g in t])
        m[0] = float(loss.detach())
        m[1] = float((torch.round(p.detach()) == yt).float().mean())

    def dp_apply(self):
        gs, o = [], 0
        for w, k in zip(self.m.ws, self.sizes):
            gs.append(self.flat[o:o + k].numpy().reshape(w.shape).copy())
            o += k
        self.m.ws = self.m.opt.step(self.m.ws, gs)


def _oracle_smallres_weights(seed=0):
    rs = np.random.RandomState(seed)
    shapes = [(3, 3, 3, 32), (32,), (3, 3, 32, 32), (32,), (3, 3, 32, 64), (64,), (3, 3, 64, 64), (64,), (2 * 2 * 64, 24), (24,),
              (24, 16), (16,), (16, 8), (8,), (8, 2), (2,)]
    return [(rs.randn(*s) * (0.2 if len(s) > 1 else 0.01)).astype(np.float32) for s in shapes]


def _smallres_dp_case():
    rs = np.random.RandomState(4)
    n = 7
    L, R = rs.rand(n, 16, 16, 3).astype(np.float32), rs.rand(n, 16, 16, 3).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[rs.randint(0, 2, n)]
    sw = np.ones(n, np.float32)
    sw[1] = 0.0
    sw[4] = 3.0
    return L, R, y, sw


def _smallres_dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import a_link_amd  # noqa: F401
        from a_link_amd import distributed as D
        L, R, y, sw = _smallres_dp_case()
        net = _OracleSmallRes(_oracle_smallres_weights())
        np.random.seed(3)
        ms = [D.dp_train_on_batch(net, [L, R], y, sample_weight=sw, mode="auto", exchange="allreduce") for _ in range(2)]   # 7 rows >= 6: sharded
        ms.append(D.dp_train_on_batch(net, [L[:2], R[:2]], y[:2], mode="sharded", exchange="host"))     # world 3: the last rank's shard is empty
        ms.append(D.dp_train_on_batch(net, [L[:5], R[:5]], y[:5], mode="auto"))                          # 5 rows < 6: replicated
        q.put((rank, np.concatenate([w.ravel() for w in net.m.ws]), np.asarray(ms, np.float64)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_smallres_dp_step_control_flow_over_gloo(world):
    """distributed.dp_train_on_batch for an end-to-end model with dropout (reference code/siamese.py:134-170):
    slices with the global normaliser, the slice's rows of the WHOLE batch's masks under rank 0's seed, one exchange of
    [gradients | metrics], the same update everywhere — over the oracle's autograd arithmetic on 2 and 3 ranks: every rank
    ends with the same weights, equal to the single-process steps to f64-level summation differences, with the same metrics;
    a batch below the model's DP_SHARD_MIN_ROWS takes the replicated form."""
    import socket
    import torch.multiprocessing as mp
    L, R, y, sw = _smallres_dp_case()
    ref = _OracleSmallRes(_oracle_smallres_weights())
    np.random.seed(3)
    want_m = [ref.train_on_batch([L, R], y, sample_weight=sw) for _ in range(2)]
    want_m.append(ref.train_on_batch([L[:2], R[:2]], y[:2]))
    want_m.append(ref.train_on_batch([L[:5], R[:5]], y[:5]))
    want_w = np.concatenate([w.ravel() for w in ref.m.ws])
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_smallres_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, w, m in got:
        assert np.array_equal(w, got[0][1]), r
        assert np.abs(w - want_w).max() < 5e-6, (r, np.abs(w - want_w).max())
        assert np.abs(m - np.asarray(want_m, np.float64)).max() < 2e-5, (r, m, want_m)      # f32 partial sums of a loss ~1.4 in a different order


# ---- rank placement: two ranks on one device must fail fast, on every rank -------------------------------
def _placement_worker(rank, world, port, q, identities):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import a_link_amd  # noqa: F401
        from a_link_amd import distributed as D
        out = []
        for ident in identities:
            try:
                D.check_rank_placement(None, identity=ident[rank])
                out.append("ok")
            except RuntimeError as e:
                out.append(str(e))
            D._PLACEMENT_CHECKED.clear()
            dist.barrier()
        # under gloo (no identity given) the check is a no-op: RowShards builds
        D.RowShards(5)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_device_raise_on_every_rank():
    """distributed.check_rank_placement: every rank publishes (host, device); two ranks on the same device make ALL ranks raise
    the same message naming them (a forgotten torch.cuda.set_device(LOCAL_RANK) otherwise shows up as an RCCL hang much
    later); distinct devices, or the same device index on different hosts, pass.  Here over gloo with faked identities."""
    import socket
    import torch.multiprocessing as mp
    world = 3
    cases = [[("hostA", 0), ("hostA", 1), ("hostA", 2)],          # fine
             [("hostA", 0), ("hostA", 1), ("hostA", 0)],          # ranks 0 and 2 share cuda:0
             [("hostA", 0), ("hostB", 0), ("hostC", 0)]]          # device 0 of three hosts: fine
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_placement_worker, args=(r, world, port, q, cases)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r][0] == "ok" and got[r][2] == "ok", got[r]
        assert "ranks [0, 2] are all on device 0 of host hostA" in got[r][1] and "set_device" in got[r][1], got[r][1]
    assert got[0][1] == got[1][1] == got[2][1]

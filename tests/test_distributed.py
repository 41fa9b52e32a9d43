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

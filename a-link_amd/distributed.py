"""distributed — one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  The reference is single-process/single-GPU (code/face_model.py:46, code/ALINK_arc.py:22-25);
this is new work shaped by SURVEY.md §8e:

  * pool inference shards by image with NO data-path collective (`shard_range`, `embed_pool_sharded`);
  * pool top-k: per-rank exact top-k, ONE all-gather of k (score, global index) candidates per rank,
    identical deterministic merge on every rank (`merge_topk`);
  * fine-tune step: each rank runs forward/backward on its slice of the batch with the GLOBAL
    normaliser, the flat gradient buffer is all-reduced (sum), every rank applies the same Adadelta
    update (`dp_train_on_batch`).
All functions work on CPU tensors too (that is how the world-size-2 gloo tests exercise them).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous [lo, hi) slice of n items for `rank`; the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _dist():
    import torch.distributed as dist
    return dist


def merge_topk(local_vals, local_global_idx, k, largest=True, group=None):
    """local_vals/local_global_idx: this rank's candidates (any length <= k, already its local top-k).
    Returns (vals, idx) of the global top-k, identical on every rank; ties -> lower global index."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = local_vals.device
    pad_v = torch.full((k,), float("-inf") if largest else float("inf"), dtype=torch.float32, device=dev)
    pad_i = torch.full((k,), np.iinfo(np.int64).max, dtype=torch.int64, device=dev)
    n = min(k, local_vals.numel())
    pad_v[:n] = local_vals[:n].to(torch.float32)
    pad_i[:n] = local_global_idx[:n].to(torch.int64)
    gv = [torch.empty_like(pad_v) for _ in range(world)]
    gi = [torch.empty_like(pad_i) for _ in range(world)]
    dist.all_gather(gv, pad_v, group=group)
    dist.all_gather(gi, pad_i, group=group)
    v = torch.cat(gv).cpu().numpy()
    i = torch.cat(gi).cpu().numpy()
    keep = i != np.iinfo(np.int64).max
    v, i = v[keep], i[keep]
    order = np.lexsort((i, -v if largest else v))[:k]       # primary: score, secondary: index
    return torch.from_numpy(v[order]).to(dev), torch.from_numpy(i[order]).to(dev)


def allreduce_sum_(t, group=None):
    _dist().all_reduce(t, group=group)
    return t


def dp_batch_slices(n, world):
    return [shard_range(n, r, world) for r in range(world)]


def dp_train_on_batch(head, x, y, class_weight=None, sample_weight=None, group=None):
    """Data-parallel Keras train_on_batch on a DenseHead: x=[L,R], y one-hot (FULL batch on every rank,
    each rank touches only its slice).  Equivalent to the single-GPU step up to f32 summation order."""
    import torch
    from . import _abi
    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = len(y)
    sw = head._sample_weights(y, class_weight, sample_weight)
    w_all = np.ones(n, np.float32) if sw is None else np.asarray(sw, np.float32)
    denom = float((w_all != 0).sum())                        # Keras: mean(w*l) / mean(w != 0)
    lo, hi = shard_range(n, rank, world)
    # gradients and {loss, accuracy-sum} travel in ONE all-reduce: the metrics live in the spare floats
    # that follow the flat gradient buffer
    gm = head.grads_tensor(with_metrics=True)
    m = gm[-4:]
    m.zero_()
    if hi > lo:
        take = lambda a: a[lo:hi] if hasattr(a, "shape") else np.asarray(a)[lo:hi]
        L, R = head._dev(take(x[0])), head._dev(take(x[1]))
        yd, swd = head._dev(take(y)), (None if sw is None else head._dev(w_all[lo:hi]))
        _abi.check(head.lib.alink_head_train_step(head.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yd), _abi.ptr(swd),
                                                  hi - lo, 1.0 / denom, 0, _abi.ptr(m), _abi.current_stream()))
        m[1] *= (hi - lo)                                    # accuracy: local mean -> local sum
    else:
        gm.zero_()
    dist.all_reduce(gm, group=group)
    _abi.check(head.lib.alink_head_apply_update(head.h, _abi.current_stream()))
    out = m[:2].cpu().numpy()
    return [float(out[0]), float(out[1] / n)]


def embed_pool_sharded(feature_model, X, group=None, gather=True):
    """Each rank embeds its contiguous shard of X (N,H,W,3); optionally all-gather the (N,512) matrix
    (C2 in SURVEY.md §2b: 100k x 512 f32 = 205 MB total)."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(len(X), rank, world)
    mine = feature_model.process(X[lo:hi])
    if not gather:
        return mine, (lo, hi)
    t = torch.as_tensor(mine)
    sizes = [shard_range(len(X), r, world) for r in range(world)]
    maxn = max(h - l for l, h in sizes)
    pad = torch.zeros((maxn, t.shape[1]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:h - l] for o, (l, h) in zip(outs, sizes)]), (lo, hi)

"""distributed — one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  The reference is single-process/single-GPU (code/face_model.py:46, code/ALINK_arc.py:22-25);
this is new work shaped by SURVEY.md §8e:

  * pool inference shards by image with NO data-path collective (`shard_range`, `embed_pool_sharded`);
  * pool top-k: per-rank exact top-k, ONE all-gather of k (score, global index) candidates per rank,
    identical deterministic merge on every rank (`merge_topk`);
  * fine-tune step (`dp_train_on_batch`): REPLICATED below DP_SHARD_MIN_ROWS rows (every rank runs the whole
    batch: at the reference's batch 16 the step is 0.04 ms and any collective costs more than it saves); above,
    each rank runs forward/backward on its slice of the batch with the GLOBAL normaliser, the flat gradient
    buffer is all-reduced (sum), every rank applies the same Adadelta update;
  * `committee_pool_topk`: the whole config-3 shape on one rank (shard -> committee of backbones + heads ->
    uncertainty -> local top-k -> merge); `committee_pool_topk_settled`: the same result from a 16-bit screening pass
    plus exact re-embedding of only the images that own a pair near the cut (settle.py).
  * `RowShards`: the rows of ONE pair batch split contiguously over the ranks — what the multi-rank A-LINK iteration
    (alink_loop.alink_iteration(group=...), BASELINE configs[3] / configs[4]) is built from: every rank perturbs, embeds
    and scores its rows, ONE all-gather carries the (P, 2) student predictions (not the features), selection is
    replicated, settle requests and the fine-tune set's rows are served by the rank that owns them.
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


def resolve_group(group):
    """The loop-level functions (alink_loop, DenseHead.dp_group) use None for "one process, no collective"; the ranks of
    a job are named by a ProcessGroup object — torch.distributed.group.WORLD for all of them, which True / "world"
    abbreviate here."""
    if group is None or group is False:
        return None
    if group is True or group == "world":
        return _dist().group.WORLD
    return group


def merge_topk(local_vals, local_global_idx, k, largest=True, group=None):
    """local_vals / local_global_idx: this rank's candidates — its local top-k, sorted (ties -> lower index), any
    length <= k.  Returns (vals, idx int64) of the global top-k, identical on every rank; ties -> lower global index.

    ONE all-gather of k (score bits, index) int32 pairs per rank (k = 1024: 8 KB per rank), then the same exact
    selection on every rank: on the GPU that is alink_topk over the world * k gathered scores.  alink_topk breaks ties
    towards the lower POSITION; ranks own contiguous, ascending index ranges (shard_range) and each rank's list is
    already tie-ordered, so position order among equal scores is global-index order.  Global indices must fit int32
    (alink_topk's own limit is P < 2^31).  CPU tensors (the gloo tests) take the same exchange and a NumPy merge."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = local_vals.device
    if k <= 0:                          # the same k on every rank: nothing to exchange
        return local_vals[:0].to(torch.float32), local_global_idx[:0].to(torch.int64)
    n = min(k, local_vals.numel())
    # preconditions (checked, not assumed): the exchange carries indices as int32, and "ties -> lower global index"
    # rests on every rank handing in its candidates best-first with ties in ascending index order, from a contiguous
    # ascending shard (what shard_range + alink_topk produce).  A rank whose candidates fail does NOT raise before the
    # collective (its peers would wait in the all-gather for ever): the failure code travels in its slot of the exchange
    # and every rank raises the same ValueError after it.
    code = 0
    if n:
        # ONE device reduction and ONE read-back for all of them (this runs once per pass of the config-3 path)
        gi = local_global_idx[:n].to(torch.int64)
        v_ = local_vals[:n].to(torch.float32)
        flags = [((gi >= 2 ** 31) | (gi < 0)).any(), torch.isnan(v_).any()]
        if n > 1:
            worse = (v_[1:] > v_[:-1]) if largest else (v_[1:] < v_[:-1])          # comparisons, not differences: equal
            flags.append((worse | ((v_[1:] == v_[:-1]) & (gi[1:] <= gi[:-1]))).any())   # infinities must count as ties
        code = int((torch.stack(flags).to(torch.int32) * torch.tensor([1, 2, 4][:len(flags)], dtype=torch.int32, device=dev)).sum())
    pad = float("-inf") if largest else float("inf")
    mine = torch.empty((k, 2), dtype=torch.int32, device=dev)
    mine[:, 0] = torch.full((k,), pad, dtype=torch.float32, device=dev).view(torch.int32)
    mine[:, 1] = -1
    if n and not code:
        mine[:n, 0] = local_vals[:n].to(torch.float32).contiguous().view(torch.int32)
        mine[:n, 1] = local_global_idx[:n].to(torch.int32)
    if code:
        mine[:, 1] = -2 - code              # index < -1: "this rank's candidates failed check `code`"

    def raise_if_any_rank_failed(idx_col):
        bad = int(idx_col.min())
        if bad < -1:
            c = -2 - bad
            who = "this rank's" if code else "another rank's"
            if c & 1:
                raise ValueError("merge_topk (%s candidates): global pair indices must lie in [0, 2^31): shard the pool into passes of fewer pairs" % who)
            if c & 2:
                raise ValueError("merge_topk (%s candidates): a candidate score is NaN (scores must be ordered: an embedding left the "
                                 "float16 range, or a head produced NaN)" % who)
            raise ValueError("merge_topk (%s candidates): candidates must be sorted best-first with ties in ascending index order" % who)
    every = torch.empty((world * k, 2), dtype=torch.int32, device=dev)
    if dev.type == "cuda":
        dist.all_gather_into_tensor(every, mine, group=group)
        raise_if_any_rank_failed(every[:, 1])
        from . import uncertainty as _unc
        vals_all = every[:, 0].contiguous().view(torch.float32)
        # padding entries (index -1) score the worst possible value; a REAL candidate with that same score (-inf / +inf)
        # ties with them, and position order could then rank padding of an earlier rank first: select among the real
        # entries only (stable compaction keeps the rank-major, tie-ordered positions alink_topk's tie rule relies on)
        real = (every[:, 1] >= 0).nonzero().flatten()
        if real.numel() == 0:
            return vals_all[:0], every[:0, 1].to(torch.int64)
        pos, v = _unc.topk_device(vals_all[real].contiguous(), min(k, int(real.numel())), largest=largest)
        idx = every[:, 1][real[pos.long()]]
        return v, idx.to(torch.int64)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    every = torch.cat(parts)
    raise_if_any_rank_failed(every[:, 1])
    every = every.numpy()
    v = every[:, 0].copy().view(np.float32)
    i = every[:, 1].astype(np.int64)
    keep = i >= 0
    v, i = v[keep], i[keep]
    order = np.lexsort((i, -v if largest else v))[:k]       # primary: score, secondary: index
    return torch.from_numpy(v[order]), torch.from_numpy(i[order])


def allreduce_sum_(t, group=None):
    _dist().all_reduce(t, group=group)
    return t


def dp_batch_slices(n, world):
    return [shard_range(n, r, world) for r in range(world)]


# Rows below which the fine-tune step is REPLICATED (every rank runs the whole batch, no communication) instead of
# sharded: one step of the 295,618-parameter head at batch 16 is three launches, 0.039 ms on one MI355X
# (DESIGN.md §4); sharding it adds a 1.18 MB all-reduce (ring over xGMI: >= 7 hops of latency, tens of
# microseconds) plus a host read-back, to save a fraction of 0.039 ms.  The kernels are deterministic, so replicas
# fed the same batch stay bit-identical.  Sharding starts to pay when a rank's slice is itself thousands of rows
# (the generic chain at 4096 rows is ~6 GFLOP of f32 MFMA work, ~60 us): the reference never gets there
# (batch 16, code/ALINK_arc.py:245), callers with large batches can force it with mode="sharded".
DP_SHARD_MIN_ROWS = 2048


def dp_train_on_batch(head, x, y, class_weight=None, sample_weight=None, group=None, mode="auto", exchange="gather"):
    """Keras train_on_batch in a one-process-per-GPU job, for a DenseHead (the siamese fine-tune step, reference
    code/siamese.py:52-58) or a SmallResNet (the end-to-end student of code/ALINK_MTP.py:121,255, code/siamese.py:134-170 —
    SURVEY.md §8e: "all-reduce(sum) of head / SmallRes gradients"): x=[L,R], y one-hot — the FULL batch on every rank.
    mode "replicated": every rank runs the whole step (bit-identical weights, zero communication);
    "sharded": each rank runs its slice with the GLOBAL normaliser (and, for SmallRes, the dropout masks of its GLOBAL
    rows), ONE exchange carries gradients + metrics, every rank applies the same Adadelta update (equal to the single-GPU
    step up to f32 summation order);
    "auto": replicated below the model's DP_SHARD_MIN_ROWS rows.
    exchange (sharded mode): "gather" = ONE all-gather of every rank's flat gradient buffer (1.18 MB head-512, 20.2 MB
    SmallRes 32 x 32 / 2048) over the direct xGMI links and a sum in rank order on every rank (one-shot, latency of one hop,
    the same bits on every rank whatever algorithm the library would pick for a reduction: SURVEY.md §5);
    "allreduce" = torch.distributed.all_reduce (RCCL's own choice); "host" = staged through the host (gloo).
    The model's side of it is four methods — grads_tensor(with_metrics=True), dp_begin, dp_local_grads, dp_apply (head.py,
    smallres.py) — so that the control flow here is what the CPU tests run over the oracle's arithmetic."""
    if mode == "auto":
        mode = "replicated" if len(y) < getattr(head, "DP_SHARD_MIN_ROWS", DP_SHARD_MIN_ROWS) else "sharded"
    if mode == "replicated":
        return head.train_on_batch(x, y, class_weight=class_weight, sample_weight=sample_weight)
    if mode != "sharded":
        raise ValueError("mode must be auto, replicated or sharded")
    import torch
    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = len(y)
    from .head import DenseHead
    sw = DenseHead._sample_weights(y, class_weight, sample_weight)
    w_all = np.ones(n, np.float32) if sw is None else np.asarray(sw, np.float32)
    denom = float((w_all != 0).sum())                        # Keras: mean(w*l) / mean(w != 0)
    lo, hi = shard_range(n, rank, world)
    # gradients and {loss, accuracy-sum} travel in ONE exchange: the metrics live in the spare floats
    # that follow the flat gradient buffer
    gm = head.grads_tensor(with_metrics=True)
    m = gm[-4:]
    m.zero_()
    ctx = head.dp_begin(n, group)                            # what every rank must agree on before its slice (SmallRes: the mask seed)
    if hi > lo:
        head.dp_local_grads(x, y, None if sw is None else w_all, lo, hi, n, 1.0 / denom, m, ctx)
        m[1] *= (hi - lo)                                    # accuracy: local mean -> local sum
    else:
        gm.zero_()
    if exchange == "gather" and world > 1 and gm.is_cuda:
        every = torch.empty((world, gm.numel()), dtype=gm.dtype, device=gm.device)
        dist.all_gather_into_tensor(every, gm, group=group)
        torch.sum(every, dim=0, out=gm)                      # fixed (rank) order
    elif exchange in ("gather", "allreduce"):
        dist.all_reduce(gm, group=group)
    elif exchange == "host":
        # a backend without device collectives (gloo): the buffer is staged through the host
        hbuf = gm.cpu()
        dist.all_reduce(hbuf, group=group)
        gm.copy_(hbuf)
    else:
        raise ValueError("exchange must be gather, allreduce or host")
    head.dp_apply()
    out = m[:2].cpu().numpy()
    return [float(out[0]), float(out[1] / n)]


def committee_pool_topk(backbones, heads, pool_shard, gallery, k, shard_offset, kind="entropy", group=None):
    """BASELINE configs[2] on one rank of a one-process-per-GPU job: this rank's pool shard
    (n, H, W, 3) and the replicated gallery are embedded by every committee member's backbone, each member's head
    scores the n * g (pool, gallery) pairs on its own embeddings, Bagging mean (reference code/committee.py:13-20),
    uncertainty measure `kind` (code/uncertainty.py), local exact top-k, ONE candidate exchange (merge_topk).
    Returns (scores, global pair index) of the job-wide top-k, identical on every rank; pair index =
    (shard_offset + i) * g + j for pool image i of this shard and gallery image j."""
    import torch
    from . import head as _head
    from . import uncertainty as _unc
    dist = _dist()
    n, g = len(pool_shard), len(gallery)
    Ep = [bb.embed_device(pool_shard) if hasattr(pool_shard, "detach") else torch.as_tensor(bb.embed(pool_shard)) for bb in backbones]
    Eg = [bb.embed_device(gallery) if hasattr(gallery, "detach") else torch.as_tensor(bb.embed(gallery)) for bb in backbones]
    dev = heads[0].device
    li = torch.arange(n, dtype=torch.int32, device=dev).repeat_interleave(g)
    ri = torch.arange(g, dtype=torch.int32, device=dev).repeat(n)
    probs = _head.committee_predict_device(heads, Ep, Eg, li, ri)
    scores = _unc.score_device(probs, kind)
    kk = min(k, scores.numel())
    idx, vals = _unc.topk_device(scores, kk, largest=True)
    gidx = idx.to(torch.int64) + int(shard_offset) * g
    if dist.is_available() and dist.is_initialized():
        return merge_topk(vals, gidx, k, largest=True, group=group)
    return vals, gidx


def merge_calibration(backbones, group=None):
    """Every rank's split-precision scales -> their elementwise MINIMUM on every rank (a scale only ever goes down when a
    batch leaves its range: the minimum covers what every rank has seen).  For the moment after some rank re-calibrated
    by itself; returns True if this rank's scales changed."""
    dist = _dist()
    mine = [bb.state() if hasattr(bb, "state") else None for bb in backbones]
    every = [None] * dist.get_world_size(group)
    dist.all_gather_object(every, mine, group=group)
    changed = False
    for i, bb in enumerate(backbones):
        if not mine[i]:
            continue
        lowest = np.min(np.asarray([st[i]["scale_exponents"] for st in every], np.int64), axis=0)
        if list(lowest) != list(mine[i]["scale_exponents"]):
            bb.load_state(dict(mine[i], scale_exponents=[int(v) for v in lowest]))
            changed = True
    return changed


def committee_pool_topk_settled(screen_backbones, exact_backbones, heads, pool_shard, gallery, k, shard_offset,
                                kind="entropy", group=None, settle_selected=True, info=None, **settle_kw):
    """_committee_pool_topk_settled under a calibration guard: the exact handles re-calibrate themselves when a
    batch leaves the split-precision range, which changes the last bits of everything embedded afterwards — the gallery
    (embedded first) and the rows settled later would then mix two sets of scales, silently.  The scales are compared
    before and after; on a change (on ANY rank: one small all-reduce) the ranks take the elementwise minimum of their
    scales and the pass runs ONCE more under them, gallery included; a second change raises.  info["recalibrated"]
    says whether that happened."""
    from . import settle as _settle
    states = lambda: [bb.state() if hasattr(bb, "state") else None for bb in exact_backbones]
    comm = _settle.make_comm(group)
    inf = {}
    for attempt in (0, 1):
        before = states()
        out = _committee_pool_topk_settled(screen_backbones, exact_backbones, heads, pool_shard, gallery, k, shard_offset,
                                           kind=kind, group=group, settle_selected=settle_selected, info=inf, **dict(settle_kw))
        changed = float(comm.sum([0.0 if states() == before else 1.0])[0]) > 0
        if not changed:
            inf["recalibrated"] = bool(attempt)
            if info is not None:
                info.update(inf)
            return out
        if comm.world > 1:
            merge_calibration(exact_backbones, group)
    raise RuntimeError("committee_pool_topk_settled: the exact backbones re-calibrated themselves in two passes in a row: "
                       "calibrate() them on images like the pool's first")


def _committee_pool_topk_settled(screen_backbones, exact_backbones, heads, pool_shard, gallery, k, shard_offset,
                                 kind="entropy", group=None, settle_selected=True, info=None, **settle_kw):
    """committee_pool_topk's result at close to the screening rate: screen-then-settle (settle.py) — on every workload
    measured, scores, order and indices equal the all-exact run's bit for bit when settle_selected (the same set otherwise),
    under an error bound that is measured and audited on a uniform sample (info["audit"]), not proven.
    `screen_backbones[m]` / `exact_backbones[m]` are member m's backbone in the 16-bit screening mode and in the exact
    mode ("f16x2" or "f32").  The gallery (replicated, a handful of images) is embedded in the exact mode only; the pool
    shard in the screening mode; then only the pool images that own a pair whose side of the k-th cut is uncertain —
    under an error bound measured on the pairs already settled, never assumed — are re-embedded in the exact mode and
    their pairs re-scored.  One candidate exchange + two small all-reduces per round (2-4 rounds).  `info` (a dict, if
    given) receives images_settled, rounds, delta, d_max, widened, fraction_re_embedded.  settle_kw: safety, delta0,
    min_sample, stage_above, max_rounds (settle.settle_topk)."""
    import torch
    from . import head as _head
    from . import settle as _settle
    from . import uncertainty as _unc
    n, g = len(pool_shard), len(gallery)
    dev = heads[0].device
    as_dev = lambda bb, x: bb.embed_device(x) if hasattr(x, "detach") else torch.as_tensor(bb.embed(x)).to(dev)
    Eg = [as_dev(bb, gallery) for bb in exact_backbones]
    Ep = [as_dev(bb, pool_shard) for bb in screen_backbones]
    li = torch.arange(n, dtype=torch.int32, device=Eg[0].device).repeat_interleave(g)
    ri = torch.arange(g, dtype=torch.int32, device=Eg[0].device).repeat(n)
    probs = _head.committee_predict_device(heads, Ep, Eg, li, ri)
    score_s = _unc.score_device(probs, kind).cpu().numpy()
    p_s = probs[:, 0].cpu().numpy()
    del Ep, probs

    def exact_fn(imgs):
        m = len(imgs)
        if hasattr(pool_shard, "detach"):
            x = pool_shard[torch.from_numpy(imgs).to(pool_shard.device)]
        else:
            x = pool_shard[imgs]
        Ex = [as_dev(bb, x) for bb in exact_backbones]
        lj = torch.arange(m, dtype=torch.int32, device=Eg[0].device).repeat_interleave(g)
        rj = torch.arange(g, dtype=torch.int32, device=Eg[0].device).repeat(m)
        pr = _head.committee_predict_device(heads, Ex, Eg, lj, rj)
        sc = _unc.score_device(pr, kind)
        pos = (imgs[:, None] * g + np.arange(g)).ravel()
        return pos, pr[:, 0].cpu().numpy(), sc.cpu().numpy()

    owner = np.repeat(np.arange(n), g)
    comm = _settle.make_comm(group)
    if "stage_above" not in settle_kw:
        settle_kw["stage_above"] = 2 * getattr(exact_backbones[0], "max_batch", 292)
    vals, gidx, inf = _settle.settle_topk(p_s, score_s, owner, n, exact_fn, k, kind=kind, largest=True, comm=comm,
                                          base=int(shard_offset) * g, settle_selected=settle_selected, **settle_kw)
    inf["fraction_re_embedded"] = inf["images_settled"] / float(max(n, 1))
    if info is not None:
        info.update(inf)
    out_dev = Eg[0].device
    return torch.from_numpy(np.ascontiguousarray(vals)).to(out_dev), torch.from_numpy(np.ascontiguousarray(gidx)).to(out_dev)


_PLACEMENT_CHECKED = set()
_PLACEMENT_CALLS = [0]           # collective call counter: a generation in the store keys (no rank reads an earlier check's entry)


def check_rank_placement(group=None, identity=None):
    """One process per GPU means ONE: a caller that forgets torch.cuda.set_device(LOCAL_RANK) puts every rank on cuda:0, and RCCL
    then fails late (or hangs) inside the first collective.  Under the "nccl" backend every rank publishes (hostname, current
    device) and ALL ranks raise the same error when two of them share a device — before any collective touches the card.  The
    identities travel through the job's rendezvous store where there is one (no communicator involved), else through
    all_gather_object.  Once per (group, world size); a no-op under "gloo" (CPU tests, two test processes sharing a card) unless
    `identity` is given."""
    dist = _dist()
    world = dist.get_world_size(group)
    if identity is None:
        if dist.get_backend(group) != "nccl":
            return
        import socket
        import torch
        identity = (socket.gethostname(), int(torch.cuda.current_device()))
    key = (id(group) if group is not None else None, world)
    if key in _PLACEMENT_CHECKED:
        return
    rank = dist.get_rank(group)
    _PLACEMENT_CALLS[0] += 1
    gen = _PLACEMENT_CALLS[0]
    every = None
    if group is None or group is getattr(dist.group, "WORLD", None):
        try:
            store = dist.distributed_c10d._get_default_store()
            store.set("alink_placement/%d/%d" % (gen, rank), "%s\t%s" % (identity[0], identity[1]))
            every = [tuple(store.get("alink_placement/%d/%d" % (gen, r)).decode().split("\t")) for r in range(world)]
        except Exception:
            every = None
    if every is None:
        every = [None] * world
        dist.all_gather_object(every, (str(identity[0]), str(identity[1])), group=group)
    seen = {}
    for r, ident in enumerate(every):
        seen.setdefault((str(ident[0]), str(ident[1])), []).append(r)
    shared = sorted((k, v) for k, v in seen.items() if len(v) > 1)
    if shared:
        raise RuntimeError("one process per GPU: " + "; ".join("ranks %s are all on device %s of host %s" % (v, k[1], k[0]) for k, v in shared)
                           + " — call torch.cuda.set_device(LOCAL_RANK) before building models or groups")
    _PLACEMENT_CHECKED.add(key)


class RowShards(object):
    """P rows (the pairs of one A-LINK mini-batch) split contiguously over the ranks of `group` (shard_range).  Every
    method is a collective: all ranks call it with the same global arguments.  Payloads are small (predictions: 8 B per
    row; fine-tune rows: a few hundred feature vectors), so they travel as ONE padded all-gather each — staged on the
    device under "nccl" (RCCL moves device buffers), on the host under "gloo"."""

    def __init__(self, P, group=None):
        import torch
        dist = _dist()
        self.torch, self.dist, self.group = torch, dist, group
        self.P = int(P)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        check_rank_placement(group)                          # nccl: two ranks on one device raise here, on every rank
        self.ranges = [shard_range(self.P, r, self.world) for r in range(self.world)]
        self.lo, self.hi = self.ranges[self.rank]
        self.device = "cuda:%d" % torch.cuda.current_device() if dist.get_backend(group) == "nccl" else "cpu"
        self._failed = None

    def fail(self, message):
        """This rank could not produce its part (its exact mode raised, say).  It must still take part in the next
        gather — its peers are on their way into it — so the failure travels THERE, in a flag row every gather carries, and
        every rank raises together instead of one raising and the others waiting for ever."""
        self._failed = str(message)

    def _gather(self, local, counts):
        """local: (counts[rank], ...) host array -> [(counts[r], ...) host array for every rank r]"""
        torch = self.torch
        local = np.ascontiguousarray(local)
        if self._failed is not None:                         # whatever was computed is not to be trusted: right shape, zeros
            local = np.zeros((counts[self.rank],) + local.shape[1:], local.dtype)
        assert local.shape[0] == counts[self.rank], (local.shape, counts, self.rank)
        m = max(counts) + 1                                  # + the flag row
        pad = np.zeros((m,) + local.shape[1:], local.dtype)
        pad[:local.shape[0]] = local
        pad[m - 1].flat[0] = 1 if self._failed is not None else 0
        mine = torch.from_numpy(pad).to(self.device)
        if self.device != "cpu":
            every = torch.empty((self.world,) + tuple(mine.shape), dtype=mine.dtype, device=self.device)
            self.dist.all_gather_into_tensor(every, mine, group=self.group)
            every = every.cpu().numpy()
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine, group=self.group)
            every = np.stack([p.numpy() for p in parts])
        bad = [r for r in range(self.world) if every[r, m - 1].flat[0] != 0]
        if bad:
            mine_msg, self._failed = self._failed, None
            raise RuntimeError("rank(s) %s failed before this exchange%s" % (bad, ": " + mine_msg if mine_msg else " (their own message says why)"))
        return [every[r, :counts[r]] for r in range(self.world)]

    def all_rows(self, local):
        """this rank's rows (hi - lo, ...) -> all P rows, in row order, on every rank"""
        return np.concatenate(self._gather(local, [h - l for l, h in self.ranges]))

    def owned(self, gidx):
        """positions (into this rank's rows) of the members of the ascending global index list `gidx` this rank owns"""
        gidx = np.asarray(gidx, np.int64)
        return gidx[(gidx >= self.lo) & (gidx < self.hi)] - self.lo

    def subsets(self, gidx_lists, local_values, row_shape, dtype=np.float32):
        """For every ascending global index list in `gidx_lists`: the rows' values (each of shape `row_shape`), in list
        order, on every rank.  local_values[i] holds the values of the members of gidx_lists[i] this rank owns (owned()
        order); all lists travel in ONE all-gather."""
        gidx_lists = [np.asarray(g, np.int64) for g in gidx_lists]
        row_shape = tuple(int(v) for v in row_shape)
        counts = [[int(((g >= l) & (g < h)).sum()) for g in gidx_lists] for l, h in self.ranges]
        flat = [np.asarray(v, dtype).reshape((-1,) + row_shape) for v in local_values]
        local = np.concatenate(flat) if flat else np.zeros((0,) + row_shape, dtype)
        per_rank = self._gather(local, [sum(c) for c in counts])
        out = []
        offs = [0] * self.world
        for i in range(len(gidx_lists)):
            parts = []
            for r in range(self.world):
                parts.append(per_rank[r][offs[r]:offs[r] + counts[r][i]])
                offs[r] += counts[r][i]
            out.append(np.concatenate(parts))                 # ranks own ascending ranges: rank order is list order
        return out

    def bcast(self, obj, src=0):
        """rank `src` (of the group)'s object on every rank"""
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, src) if self.group is not None else src,
                                        group=self.group)
        return box[0]

    def all_true(self, flag):
        """True on every rank iff `flag` is true on all ranks"""
        every = [None] * self.world
        self.dist.all_gather_object(every, bool(flag), group=self.group)
        return all(every)

    def same_everywhere(self, obj):
        """True on every rank iff `obj` (small, picklable) is equal on all ranks"""
        every = [None] * self.world
        self.dist.all_gather_object(every, obj, group=self.group)
        return all(e == every[0] for e in every)


def broadcast_calibration(backbones, src=0, group=None):
    """Make the split-precision calibration state of rank `src` the state of every rank (IRBackbone / VGGResNet50
    objects; other dtypes carry none and are skipped).  A rank calibrates on the images IT sees (its shard); the scales it
    picks decide the last bits of every embedding, and the job-wide top-k merge assumes the same arithmetic on every rank
    — so calibrate on one rank (or on a sample every rank draws identically), then call this; call it again after any
    re-calibration (IRBackbone re-calibrates by itself when a batch leaves the range: check `state()` for a change).  One
    small object broadcast per call."""
    dist = _dist()
    states = [bb.state() for bb in backbones] if dist.get_rank(group) == src else [None] * len(backbones)
    dist.broadcast_object_list(states, src=src, group=group)
    for bb, st in zip(backbones, states):
        bb.load_state(st)
    return states


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

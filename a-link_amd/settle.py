"""settle — screen-then-settle: the exact arithmetic's selection sets at close to the screening rate — identical on every
workload measured, under an error bound that is MEASURED (never assumed) and guarded by a sampled audit; not a theorem.

The reference makes every selection from float32 probabilities (code/face_model.py:86-93 embeds in float32) with a
handful of threshold / rank cuts: the committee's top-n most uncertain pairs (code/uncertainty.py:133-217 through
code/committee.py:13-20 and code/existing_al.py:104-110) and the A-LINK rule — per noise the int(P * ratio) largest
|M2 - M1|, the grey band 0.5 +- eps, the 0.5 decision (code/ALINK_arc.py:167-198, code/ALINK.py:170-201).  The build's
exact mode (split precision, "f16x2") reproduces those sets bit for bit but costs three matrix-core products per
multiplication; its 16-bit modes are three times faster and move a probability by up to ~2e-3 (f16) / ~2e-2 (bf16),
which turns over every pair that sits that close to a cut.  Only those pairs need the exact arithmetic:

    1. SCREEN   embed everything in the 16-bit mode, score every pair;
    2. BOUND    a pair's exact probability lies within +-delta of its screened one.  delta is not assumed: it is
                `safety` x the LARGEST |exact - screened| seen on any pair settled so far (a mandatory first sample
                of the pairs nearest the cut starts it), so a band chosen too narrow widens itself;
    3. RESOLVE  with every unsettled pair an INTERVAL and every settled pair a point, a pair's side of a rank cut is
                certain unless its interval reaches across the cut (topk_undetermined: interval arithmetic, no
                probability argument inside the band);
    4. SETTLE   re-embed in the exact mode ONLY the images that own an uncertain pair, nearest the cut first (once the
                pairs around the cut are exact the cut itself is, and the band that remains is one delta wide, not
                two), re-score those pairs, repeat from 2 until nothing is uncertain.

    5. AUDIT    the bound of step 2 is a sample maximum over the pairs NEAREST the cut; a pair far from the cut whose own
                error exceeds both delta and its distance to the cut would keep the wrong side unseen.  So once nothing is
                uncertain, a seeded UNIFORM sample of m never-settled units (images / (pair, noise) rows) is settled too:
                an error above the largest seen so far widens delta and re-opens the resolution (and the audit), until a
                sample shows none.  What that proves is statistical: no exceedance in m samples => the fraction of
                unsettled units whose error exceeds the largest error seen lies below 3 / m at 95 % confidence (rule of
                three; m = 300: below 1 %).  A single outlier among thousands can still go unseen (the CPU tests plant
                one and show both outcomes); `info["audit"]` reports m, the largest audited error and the exceedances.

What comes out is the exact path's result under that bound: members of the selected set are either settled (their exact
scores rank them, with the exact path's own tie rule) or certain by interval.  With settle_selected=True every member is
settled, and — on every workload measured (tests/, bench.py compare in the run) — scores, order and indices equal the
all-exact run bit for bit (the exact kernels are batch-invariant: an image embeds and a pair scores to the same bits
whatever batch it arrives in).

Everything here is host logic on NumPy arrays (P is 200 k pairs per GPU at BASELINE configs[2]); the arithmetic — the
two embedding modes, the pair heads, the uncertainty scores — runs in libalink_hip.so behind the callbacks the callers
pass (distributed.committee_pool_topk_settled, alink_loop.alink_iteration).  Multi-GPU: every rank resolves its own
shard; a round costs one candidate exchange (distributed.merge_topk) and two small all-reduces.
"""
import numpy as np

_NEG = np.float32(-np.inf)
_POS = np.float32(np.inf)

# the device's uncertainty score (alink_score, float32 on a float32 softmax) against the float64 expression the intervals
# are computed with: < 1e-6 (asserted in tests/test_gpu_pool.py); intervals are widened by this much in score space
SCORE_GUARD = 4e-6


def _down32(x):
    """float64 -> the largest float32 <= x"""
    x = np.asarray(x, np.float64)
    y = x.astype(np.float32)
    return np.where(y.astype(np.float64) > x, np.nextafter(y, _NEG), y).astype(np.float32)


def _up32(x):
    """float64 -> the smallest float32 >= x"""
    x = np.asarray(x, np.float64)
    y = x.astype(np.float32)
    return np.where(y.astype(np.float64) < x, np.nextafter(y, _POS), y).astype(np.float32)


def _score_of_u(u, kind):
    """The three measures of reference code/uncertainty.py:15-60 for a TWO-class row (p, 1 - p), as a function of
    u = |p - 1/2|: every one is monotone in u (A-LINK's pair scorers are two-class softmaxes, code/siamese.py:31-33)."""
    u = np.clip(np.asarray(u, np.float64), 0.0, 0.5)
    if kind == "uncertainty":
        return 0.5 - u
    if kind == "margin":
        return 2.0 * u
    if kind == "entropy":
        p, q = 0.5 + u, 0.5 - u
        with np.errstate(divide="ignore", invalid="ignore"):
            return -(p * np.log(p) + np.where(q > 0, q * np.log(np.where(q > 0, q, 1.0)), 0.0))
    raise ValueError("kind must be uncertainty, margin or entropy")


def binary_score_interval(p0, delta, kind):
    """[lo, hi] (float32, rounded outwards) of the score of every two-class row whose first probability lies within
    +-delta of p0."""
    p0 = np.asarray(p0, np.float64)
    u = np.abs(p0 - 0.5)
    s_near = _score_of_u(np.maximum(u - delta, 0.0), kind)          # nearest 1/2 the true value can be
    s_far = _score_of_u(np.minimum(u + delta, 0.5), kind)
    return _down32(np.minimum(s_near, s_far) - SCORE_GUARD), _up32(np.maximum(s_near, s_far) + SCORE_GUARD)


class LocalComm(object):
    """One process: nothing to exchange."""
    world = 1

    def topk(self, vals, gidx, k, largest=True):
        return vals, gidx

    def max(self, x):
        return np.asarray(x, np.float64)

    def sum(self, x):
        return np.asarray(x, np.float64)


class DistComm(object):
    """One process per GPU (torch.distributed; "nccl" = RCCL on the GPU box, "gloo" in the CPU tests): the candidate
    exchange is distributed.merge_topk, the scalars travel in one small all-reduce each."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        if device is None:
            from . import distributed as _D
            _D.check_rank_placement(group)                   # nccl: two ranks on one device raise here, on every rank
            device = "cuda:%d" % torch.cuda.current_device() if dist.get_backend(group) == "nccl" else "cpu"
        self.device = device

    def topk(self, vals, gidx, k, largest=True):
        from . import distributed as D
        t = self.torch
        v, i = D.merge_topk(t.from_numpy(np.ascontiguousarray(vals, np.float32)).to(self.device),
                            t.from_numpy(np.ascontiguousarray(gidx, np.int64)).to(self.device), k, largest=largest,
                            group=self.group)
        return v.cpu().numpy(), i.cpu().numpy()

    def _reduce(self, x, op):
        t = self.torch.from_numpy(np.atleast_1d(np.asarray(x, np.float64)).copy()).to(self.device)
        self.dist.all_reduce(t, op=op, group=self.group)
        return t.cpu().numpy()

    def max(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX)

    def sum(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM)


def make_comm(group=None):
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return DistComm(group)
    except ImportError:
        pass
    return LocalComm()


def _local_topk(pess, k):
    """positions of the k largest, best first, ties -> lower position"""
    P = len(pess)
    if k >= P:
        cand = np.arange(P)
    else:
        thr = np.partition(pess, P - k)[P - k]
        cand = np.flatnonzero(pess >= thr)
    return cand[np.lexsort((cand, -pess[cand].astype(np.float64)))][:k]


def topk_undetermined(lo, hi, k, largest=True, comm=None, base=0):
    """Items known as intervals [lo[i], hi[i]] (float32; lo == hi where the value is exact).  Which of them does the
    exact top-k (ties -> lower index, like alink_topk / a stable argsort) certainly hold, certainly not hold, and
    which are undetermined?

    T = the k best by the PESSIMISTIC bound.  With a = the worst pessimistic bound inside T and b = the best optimistic
    bound outside it: an interval item of T whose pessimistic bound beats b beats everything outside T in every
    realisation (at most k - 1 items can rank above it: certainly selected — that holds for an exact member of T too);
    an item outside T whose optimistic bound stays below a loses to all k members of T (certainly not selected — an
    exact item outside T always is: every member's pessimistic bound already ranks above it); exact items rank among
    each other by value and index like the exact path.  Returns (in_T, need, undetermined, a, b): `undetermined` marks
    every item whose own side is not certain yet (an exact member of T that an interval outside T can still overtake is
    one), `need` = its interval members, the ones settling can help.  When `need` has no member ANYWHERE (all ranks), T
    is the exact top-k.  Across ranks (`comm`, global index = base + position) T is global."""
    comm = comm or LocalComm()
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    if not largest:
        lo, hi = -hi, -lo
    P = len(lo)
    in_T = np.zeros(P, bool)
    if k <= 0:                  # an empty selection: nothing is in it, nothing can enter it, nothing to settle (on any rank)
        return in_T, in_T.copy(), in_T.copy(), float("inf"), float("-inf")
    order = _local_topk(lo, min(k, P)) if k > 0 else np.zeros(0, np.int64)
    gv, gi = comm.topk(lo[order], order.astype(np.int64) + int(base), k, largest=True)
    mine = gi[(gi >= base) & (gi < base + P)] - base
    in_T[mine] = True
    a = float(gv[-1]) if len(gv) >= k and k > 0 else float("-inf")        # fewer than k items in all: every one is in
    b_loc = float(hi[~in_T].max()) if (~in_T).any() else float("-inf")
    b = float(comm.max([b_loc])[0])
    und = (in_T & (lo <= np.float32(b))) | (~in_T & (hi > lo) & (hi >= np.float32(a)))
    return in_T, und & (hi > lo), und, a, b


class ErrorBound(object):
    """delta = max(delta0, safety x the largest |exact - screened| any settled pair has shown): never assumed, only
    ever widened by what was measured."""

    def __init__(self, delta0=0.0, safety=1.5):
        self.delta0, self.safety = float(delta0), float(safety)
        self.d_max = 0.0
        self.widened = 0
        self.delta = float(delta0)

    def observe(self, d_max_global):
        self.d_max = max(self.d_max, float(d_max_global))
        new = max(self.delta0, self.safety * self.d_max)
        if new > self.delta:
            if self.delta > 0:                  # a claim (delta0, or what earlier samples showed) turned out too narrow
                self.widened += 1
            self.delta = new
        return self.delta


AUDIT_DEFAULT = None        # audit sample size: None = auto (below); an int = that many units; 0 = no audit


def audit_size(audit, unsettled):
    """Units of one audit pass.  Auto: 2 % of the never-settled units, at least 32 and at most 300 (no exceedance in 300
    samples => fewer than 1 % of the unsettled lie beyond the bound, at 95 %): a cost of a few per cent of the screening pass
    whatever the workload's size."""
    if audit is None:
        return int(min(300, max(32, np.ceil(0.02 * unsettled))))
    return int(audit)


def _audit_pick(candidates, m, seed, salt):
    """m of `candidates` (ascending positions), uniformly without replacement, from a stream of its own"""
    if m <= 0 or len(candidates) == 0:
        return np.zeros(0, np.int64)
    rs = np.random.RandomState([int(seed) & 0x7FFFFFFF, int(salt) & 0x7FFFFFFF, 0x5E771E])
    return np.sort(rs.choice(candidates, min(int(m), len(candidates)), replace=False))


def settle_topk(p_screen, score_screen, owner, n_owner, exact_fn, k, kind="entropy", largest=True, comm=None, base=0,
                safety=1.5, delta0=0.0, min_sample=64, stage_above=584, settle_selected=True, max_rounds=24,
                audit=AUDIT_DEFAULT, audit_seed=0):
    """The exact top-k of `kind` over this rank's pairs, from screened probabilities.

    p_screen      (P,) first-column probability of every pair from the SCREENING embeddings
    score_screen  (P,) float32: the device's own score of the screened rows (kept for members that are certain unsettled)
    owner         (P,) int: the image that owns each pair — settling is per image (all of its pairs at once)
    exact_fn      f(sorted image indices) -> (pair positions, exact p0, exact float32 score): re-embeds those images in
                  the exact mode and re-scores every pair they own
    audit         images of the uniform audit sample per pass, over all ranks (None: audit_size()'s choice; 0: no audit — the
                  bound then rests on the pairs nearest the cut alone); audit_seed: its stream
    Returns (vals float32, global index int64, info).  info: images settled, rounds, delta, d_max, widened, audit.
    An error on one rank (the exact mode returning a non-finite value, exact_fn raising) is raised on EVERY rank: the flag
    travels in the round's all-reduce, so no rank is left waiting in a collective.
    """
    comm = comm or LocalComm()
    if k <= 0:
        return (np.zeros(0, np.float32), np.zeros(0, np.int64),
                {"images": int(n_owner), "images_settled": 0, "pairs": int(len(p_screen)), "rounds": 0, "delta": float(delta0),
                 "d_max": 0.0, "widened": 0, "members_unsettled": 0, "audit": None})
    p_screen = np.asarray(p_screen, np.float64)
    P = len(p_screen)
    owner = np.asarray(owner, np.int64)
    val = np.array(score_screen, np.float32, copy=True)
    pair_settled = np.zeros(P, bool)
    img_settled = np.zeros(int(n_owner), bool)
    bound = ErrorBound(delta0, safety)
    rounds = 0
    aud = {"m": 0, "passes": 0, "max_err": 0.0, "exceedances": 0}
    # a screened probability that is not finite (a 16-bit forward that left its range) says nothing: such a pair can be anywhere
    unknown = ~np.isfinite(p_screen)
    p_screen = np.where(unknown, 0.5, p_screen)

    def settle_images(imgs, thr=None):
        """exact_fn on `imgs` (may be empty); returns (largest |exact - screened| among them, error text or None, number
        of these images with a pair whose error exceeds `thr`)"""
        if not len(imgs):
            return 0.0, None, 0
        try:
            pos, p_x, s_x = exact_fn(imgs)
            pos = np.asarray(pos, np.int64)
            p_x, s_x = np.asarray(p_x, np.float64), np.asarray(s_x, np.float32)
            if not (np.isfinite(p_x).all() and np.isfinite(s_x).all()):
                return 0.0, "the exact mode returned a non-finite probability or score", 0
        except Exception as e:              # raised together on every rank below
            return 0.0, "%s: %s" % (type(e).__name__, e), 0
        known = ~unknown[pos]
        dd = np.where(known, np.abs(p_x - p_screen[pos]), 0.0)
        d = float(dd.max()) if len(dd) else 0.0
        over = int(len(np.unique(owner[pos][dd > thr]))) if thr is not None else 0
        val[pos] = s_x
        pair_settled[pos] = True
        img_settled[imgs] = True
        return d, None, over

    def agree(d_loc, err):
        """one all-reduce: the largest error seen on any rank + whether any rank failed"""
        got = comm.max([d_loc, 1.0 if err else 0.0])
        if got[1] > 0:
            raise RuntimeError("settle_topk: %s" % (err or "another rank failed while settling (its own message says why)"))
        return float(got[0])

    while True:
        lo, hi = binary_score_interval(p_screen, bound.delta, kind)
        lo = np.where(unknown, _NEG, lo)
        hi = np.where(unknown, _POS, hi)
        lo = np.where(pair_settled, val, lo)
        hi = np.where(pair_settled, val, hi)
        in_T, need, _, a, b = topk_undetermined(lo, hi, k, largest, comm, base)
        if settle_selected:
            need = need | (in_T & ~pair_settled)
        mid = 0.5 * (a + b) if np.isfinite(a) and np.isfinite(b) else (a if np.isfinite(a) else (b if np.isfinite(b) else 0.0))
        with np.errstate(invalid="ignore"):          # an unknown pair is (-inf, +inf): its distance to the cut counts as 0
            centre = 0.5 * (lo.astype(np.float64) + hi.astype(np.float64))
        if not largest:
            centre = -centre
        dist = np.where(np.isfinite(centre), np.abs(centre - mid), 0.0)
        imgs = np.unique(owner[need])
        if rounds == 0 and len(imgs) < min(min_sample, int((~img_settled).sum())):
            # the mandatory first sample: the images owning the pairs nearest the cut, whatever delta0 claims
            cand = np.flatnonzero(~pair_settled)
            near = cand[np.argsort(dist[cand], kind="stable")]
            extra = []
            seen = set(imgs.tolist())
            for o in owner[near]:
                if o not in seen:
                    seen.add(int(o))
                    extra.append(int(o))
                    if len(seen) >= min_sample:
                        break
            imgs = np.unique(np.concatenate([imgs, np.asarray(extra, np.int64)]))
        elif len(imgs) > stage_above:
            # nearest the cut first: once those are exact the cut is, and what is left of the band is one delta wide
            dimg = np.full(int(n_owner), np.inf)
            np.minimum.at(dimg, owner[need], dist[need])
            imgs = imgs[np.argsort(dimg[imgs], kind="stable")][:max(stage_above, (len(imgs) + 1) // 2)]
            imgs = np.sort(imgs)
        if rounds == 0 and (audit is None or audit > 0):
            # with the audit on, the first round also takes a uniform draw of a third of an audit pass: the images nearest the
            # cut under-estimate the largest error of the pool, and every later widening costs a round
            first = _audit_pick(np.flatnonzero(~img_settled), max(4, audit_size(audit, n_owner) // 3), audit_seed, int(base) + 104729)
            imgs = np.unique(np.concatenate([imgs, first]))
        todo = float(comm.sum([len(imgs)])[0])
        if todo == 0:
            # nothing is uncertain under the bound measured so far — on the pairs nearest the cut.  AUDIT it on a uniform
            # sample of the images never settled; an error above the largest seen widens the bound and re-opens the resolution
            if audit is not None and audit <= 0:
                break
            cand = np.flatnonzero(~img_settled)
            left = float(comm.sum([len(cand)])[0])
            if left == 0:
                break
            share = int(np.ceil(audit_size(audit, left) * len(cand) / left)) if len(cand) else 0       # this rank's part of the sample
            pick = _audit_pick(cand, share, audit_seed, int(base) + 7919 * aud["passes"])
            d_loc, err, over = settle_images(pick, thr=bound.d_max)
            d_aud = agree(d_loc, err)
            tot = comm.sum([len(pick), over])
            aud["m"] += int(tot[0])
            aud["passes"] += 1
            aud["max_err"] = max(aud["max_err"], d_aud)
            aud["exceedances"] += int(tot[1])
            held = bound.delta
            bound.observe(d_aud)
            rounds += 1
            if bound.delta == held:
                break                   # the sample showed nothing beyond what the bound already covers: the determination stands
            if rounds >= max_rounds:
                raise RuntimeError("settle_topk: the audit kept widening the bound (%d rounds, delta %.3g)" % (rounds, bound.delta))
            continue
        if rounds >= max_rounds:
            raise RuntimeError("settle_topk: %d images still undetermined after %d rounds (delta %.3g)" % (todo, rounds, bound.delta))
        d_loc, err, _ = settle_images(imgs)
        bound.observe(agree(d_loc, err))
        rounds += 1
    # every member of T is settled or certain; non-members never outrank one.  Hand the candidate exchange T itself.
    cand = np.flatnonzero(in_T)
    key = val[cand] if largest else -val[cand]
    order = cand[np.lexsort((cand, -key.astype(np.float64)))]
    vals, gidx = comm.topk(val[order], order.astype(np.int64) + int(base), k, largest=largest)
    info = {"images": int(n_owner), "images_settled": int(img_settled.sum()), "pairs": int(P), "rounds": rounds,
            "delta": bound.delta, "d_max": bound.d_max, "widened": bound.widened,
            "members_unsettled": int((in_T & ~pair_settled).sum()),
            "audit": dict(aud, unit="image", claim="no exceedance in m uniform samples => fraction of never-settled images whose error "
                                                   "exceeds the largest error seen < 3/m at 95 %") if (audit is None or audit > 0) else None}
    return vals, gidx, info


# ---------------------------------------------------------------------------------------------------------------------
# The A-LINK rule (reference code/ALINK_arc.py:167-198 column 0, code/ALINK.py:170-201 column 1) from screened noisy passes
# ---------------------------------------------------------------------------------------------------------------------
DIFF_GUARD = 2e-7        # |M2 - M1| is one float32 subtraction of two float32 probabilities


def _disparity_interval(m, e, delta):
    d = np.abs(np.asarray(m, np.float64) - np.asarray(e, np.float64))
    return _down32(np.maximum(d - delta - DIFF_GUARD, 0.0)), _up32(d + delta + DIFF_GUARD)


def select_queries_settled(ensemblePredictions, disguisedScreened, batch_y, settle_fn, col=0, disparity_ratio=0.25,
                           eps=0.05, blind_strategy=False, safety=1.5, delta0=0.0, min_sample=32, max_rounds=24,
                           settle_many=None, audit=AUDIT_DEFAULT, audit_seed=0):
    """selection.select_queries with the noisy passes SCREENED: the clean pass (`ensemblePredictions`, every unique image
    of the mini-batch once: a few dozen embeddings) is exact, the 2 P n_noise noisy pair occurrences — the bulk of an
    iteration's embeddings — were embedded in the 16-bit mode, and `settle_fn(k, pairs)` returns the
    student's EXACT predictions for those pairs' noise-k copies (re-embedding their 2 images each in the exact mode).
    A (pair, noise) is settled only when the pair's side of that noise's rank cut (or, blind strategy, of the 0.5
    decision) is uncertain AND the pair can still reach the query set (outside the grey band — the clean pass is exact,
    so that is known — and not certainly cut by another noise).  Afterwards the selected pairs' assigned noise copies
    (chunk i of the query list takes noise i, code/ALINK_arc.py:213-222) are settled too: what goes to the fine-tune set
    is exact.  Returns (queryIndices ascending, active_count, labels, disguisedPredictions with the settled rows exact,
    settled masks per noise, info) — queryIndices / active_count / labels equal selection.select_queries on all-exact
    predictions.  settle_many (optional): f([(k, pairs), ...]) -> [predictions, ...] — a whole round's requests in ONE
    call, so that the caller can embed them as one batch (a round asks for a few dozen to a few hundred pairs per noise).
    audit / audit_seed: once nothing is uncertain, a seeded uniform sample of never-settled (pair, noise) rows is settled
    as well (module header, step 5); an error above the largest seen widens the bound and re-opens the resolution.
    info["audit"] = {m, passes, max_err, exceedances}."""
    from .helpers import roundoff
    ens = np.asarray(ensemblePredictions)
    P = len(ens)
    n_noise = len(disguisedScreened)
    e = ens[:, col].astype(np.float32)
    dis = [np.array(d, np.float32, copy=True) for d in disguisedScreened]
    for d in dis:                                          # placeholders for unknown rows until they are settled
        d[~np.isfinite(d).all(axis=1)] = 0.5
    scr = [np.asarray(d)[:, col].astype(np.float64) for d in disguisedScreened]
    unknown = [~np.isfinite(v) for v in scr]               # a non-finite screened prediction says nothing: such a row can be anywhere
    scr = [np.where(u, 0.5, v) for u, v in zip(unknown, scr)]
    settled = [np.zeros(P, bool) for _ in range(n_noise)]
    notgrey = (e <= 0.5 - eps) | (e >= 0.5 + eps)
    K = int(P * disparity_ratio)
    bound = ErrorBound(delta0, safety)
    rounds = 0
    n_settled = 0

    state = {"d": 0.0}
    aud = {"m": 0, "passes": 0, "max_err": 0.0, "exceedances": 0}

    def settle_all(requests):
        """requests: [(k, pair indices)] — drop what is settled already, ask for the rest in one call.  Returns the largest
        |exact - screened| among THESE rows (state["d"] keeps the largest ever)."""
        nonlocal n_settled
        worst = 0.0
        todo = []
        for k, idx in requests:
            idx = np.asarray(sorted(set(int(i) for i in idx)), np.int64)
            idx = idx[~settled[k][idx]] if len(idx) else idx
            if len(idx):
                todo.append((k, idx))
        if not todo:
            return worst
        got = settle_many(todo) if settle_many is not None else [settle_fn(k, idx) for k, idx in todo]
        for (k, idx), px in zip(todo, got):
            px = np.asarray(px, np.float32)
            if not np.isfinite(px).all():
                raise RuntimeError("select_queries_settled: the exact mode returned a non-finite prediction")
            known = ~unknown[k][idx]
            if known.any():
                dd = np.abs(px[known, col].astype(np.float64) - scr[k][idx][known])
                worst = max(worst, float(dd.max()))
            dis[k][idx] = px
            settled[k][idx] = True
            n_settled += len(idx)
        state["d"] = max(state["d"], worst)
        return worst

    def membership():
        """per noise: (member mask by the pessimistic rule, pairs settling can help, pairs whose own side is uncertain,
        [lo, hi] of the ranked quantity, distance of every pair to that cut)"""
        out = []
        for k in range(n_noise):
            m = dis[k][:, col]
            if blind_strategy:
                mem = (m >= 0.5) != (e >= 0.5)
                und = ~settled[k] & ((np.abs(scr[k] - 0.5) <= bound.delta + DIFF_GUARD) | unknown[k])
                out.append((mem, und, und, None, None, np.abs(scr[k] - 0.5)))
                continue
            d_exact = np.abs(m - e)                                   # float32, the exact path's own expression
            lo, hi = _disparity_interval(scr[k], e, bound.delta)
            lo, hi = np.where(unknown[k], np.float32(0), lo), np.where(unknown[k], _POS, hi)
            lo = np.where(settled[k], d_exact, lo)
            hi = np.where(settled[k], d_exact, hi)
            in_T, need, und, a, b = topk_undetermined(lo, hi, K, largest=True)       # K == 0: nothing selected, nothing to settle
            mid = 0.5 * (a + b) if np.isfinite(a) and np.isfinite(b) else 0.0
            with np.errstate(invalid="ignore"):
                centre = 0.5 * (lo.astype(np.float64) + hi)
            out.append((in_T, need, und, lo, hi, np.where(np.isfinite(centre), np.abs(centre - mid), 0.0)))
        return out

    while True:
        while True:
            mem = membership()
            # a pair outside T whose own side is certain is certainly cut by that noise (valid at any time)
            certainly_out = [~m[0] & ~m[2] for m in mem]
            todo, requests = 0, []
            for k in range(n_noise):
                in_T, need, und, lo, hi, dist = mem[k]
                alive = notgrey.copy()                                # pairs that can still reach the query set
                for k2 in range(n_noise):
                    if k2 != k:
                        alive &= ~certainly_out[k2]
                ru = und & alive
                if blind_strategy:
                    idx = np.flatnonzero(ru)                          # a threshold: every pair stands alone
                elif ru.any():
                    # a rank cut: an uncertain pair that matters is decided by EVERY interval that overlaps it, whether
                    # or not that one matters itself — settle the uncertain intervals over the hull of those that do
                    h_lo, h_hi = lo[ru].min(), hi[ru].max()
                    idx = np.flatnonzero(need & (hi >= h_lo) & (lo <= h_hi))
                else:
                    idx = np.zeros(0, np.int64)
                if rounds == 0:
                    # the mandatory first sample: the pairs nearest this noise's cut, whatever delta0 claims ...
                    want = max(8, min_sample // max(n_noise, 1))
                    near = np.argsort(np.where(settled[k], np.inf, dist), kind="stable")[:want]
                    idx = np.union1d(idx, near[~settled[k][near]])
                    # ... and, with the audit on, a uniform draw of a third of an audit pass: the errors of the rows nearest a
                    # cut under-estimate the largest error of the batch (measured: 8e-4 against 2.2e-3 at config 4), and every
                    # later widening costs a resolution round — start from a bound the whole population has had a say in
                    if audit is None or audit > 0:
                        first = _audit_pick(np.arange(P), max(4, audit_size(audit, P * n_noise) // (3 * max(n_noise, 1))), audit_seed, 104729 + k)
                        idx = np.union1d(idx, first)
                todo += len(idx)
                requests.append((k, idx))
            settle_all(requests)
            bound.observe(state["d"])
            rounds += 1
            if todo == 0:
                break
            if rounds > max_rounds:
                raise RuntimeError("select_queries_settled: still undetermined after %d rounds (delta %.3g)" % (rounds, bound.delta))
        works = np.ones(P, bool)
        for m in mem:
            works &= m[0]
        queryIndices, active = [], 0
        for j in np.flatnonzero(works & notgrey):
            active += 1
            if (e[j] >= 0.5) == (batch_y[j][0] >= 0.5):
                queryIndices.append(int(j))
        # what the fine-tune set takes from the noisy passes: chunk i of the query list <- noise i
        mp = int(len(queryIndices) / float(n_noise)) if n_noise else 0
        sel = [(i, np.asarray(queryIndices[i * mp:(i + 1) * mp], np.int64)) for i in range(n_noise)]
        # The determination stands under the bound measured so far — on the rows nearest the cuts.  Two more groups of rows are
        # settled, in ONE exact call (a round is an embedding call and, across ranks, two collectives): the selected rows'
        # assigned noise copies, and the AUDIT's uniform sample of the (pair, noise) rows never settled (module header, step 5)
        held = bound.delta
        audit_on = not ((audit is not None and audit <= 0) or n_noise == 0)
        pick_by_noise = [np.zeros(0, np.int64)] * n_noise
        if audit_on:
            taken = np.concatenate(settled).copy()                                   # position k * P + j
            for i, idx in sel:
                taken[i * P + idx] = True                                            # about to be settled as selected rows
            flat = np.flatnonzero(~taken)
            if len(flat):
                pick = _audit_pick(flat, audit_size(audit, len(flat)), audit_seed, 7919 * aud["passes"] + P)
                pick_by_noise = [pick[(pick >= k * P) & (pick < (k + 1) * P)] - k * P for k in range(n_noise)]
        n_pick = int(sum(len(v) for v in pick_by_noise))
        before = n_settled
        d_before = bound.d_max
        settle_all([(k, np.union1d(sel[k][1], pick_by_noise[k])) for k in range(n_noise)])
        if n_settled == before and n_pick == 0:
            break             # nothing was left to settle: points only replaced intervals, the determination stands
        rounds += 1
        if n_pick:
            # the audited rows' own errors (their exact predictions now sit in `dis`)
            errs = np.concatenate([np.abs(dis[k][v, col].astype(np.float64) - scr[k][v])[~unknown[k][v]] for k, v in enumerate(pick_by_noise)])
            aud["m"] += n_pick
            aud["passes"] += 1
            aud["max_err"] = max(aud["max_err"], float(errs.max()) if len(errs) else 0.0)
            aud["exceedances"] += int((errs > d_before).sum())
        if bound.observe(state["d"]) == held:
            break             # neither the selected rows nor the sample showed anything beyond what the bound covers
        if rounds > max_rounds:
            raise RuntimeError("select_queries_settled: the audit kept widening the bound (%d rounds, delta %.3g)" % (rounds, bound.delta))
    labels = roundoff(ens[queryIndices, col]) if queryIndices else np.zeros((0, 1), dtype=int)
    info = {"pairs": P, "noises": n_noise, "pair_noise_settled": int(n_settled), "fraction_settled": n_settled / float(max(P * n_noise, 1)),
            "rounds": rounds, "delta": bound.delta, "d_max": bound.d_max, "widened": bound.widened,
            "audit": dict(aud, unit="(pair, noise) row", claim="no exceedance in m uniform samples => fraction of never-settled rows whose "
                                                                "error exceeds the largest error seen < 3/m at 95 %")
                     if (audit is None or audit > 0) else None}
    return queryIndices, active, labels, dis, settled, info

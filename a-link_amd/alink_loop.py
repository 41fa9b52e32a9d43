"""alink_loop — the A-LINK / A2-LINK framework loop of the reference's three drivers as library code.

    code/ALINK_arc.py:62-260   ArcFace-112 teacher features (512-d), column 0
    code/ALINK.py:65-265       VGGFace2-ResNet50 teacher features (2048-d), column 1
    code/ALINK_MTP.py:78-289   high-res teacher features, low-res SmallRes student trained on pixels

The reference keeps this loop inline under `if __name__ == "__main__":` with tf.flags; here it is a
function over duck-typed models (the same objects the drivers build: a feature model with
`.process`, a committee.Bagging, noise objects, a student with `.predict/.finetune/.save`) so that it
can be driven by tests, the synthetic benchmark and real data alike.  Flags keep the reference's
names and defaults (`Flags`, `add_flags(argparse parser)`).

What changes is only where arithmetic runs and how often:
  * the clean pass embeds every UNIQUE image of the mini-batch once and gathers pairs by index
    (pairs.createMiniBatchIndices) — the reference embeds 2P pair occurrences at batch 1
    (code/ALINK_arc.py:154).  Embeddings are per-image deterministic, so the
    features are identical to embedding every occurrence;
  * noisy copies are generated per pair occurrence (independent draws, as code/noise.py:20-30 does)
    but each noise is one kernel launch and each noisy batch one batched embed;
  * selection (code/ALINK_arc.py:167-198) is selection.select_queries: same rule, ties towards the
    lower index, queryIndices ascending (Python-2 `Set` order is arbitrary in the reference).
`augment=True` (tf.contrib rotations + imgaug, code/helpers.py:114-141) is outside the hot path and
raises NotImplementedError.
"""
import numpy as np

from . import helpers, pairs, selection


class Flags(object):
    """tf.flags of the drivers (code/ALINK_arc.py:35-60), same names and defaults."""
    out_model = 'ARCFace_models/postALINK'
    ensemble_basepath = 'ARCFace_models/ensemble'
    disguised_basemodel = 'ARCFace_models/disguisedModel'
    noise = 'gaussian,saltpepper,poisson,perlin,speckle,adversarial'
    ft_epochs = 3
    batch_size = 16
    dig_epochs = 40
    undig_epochs = 60
    batch_send = 64
    mixture_ratio = 2
    alink_bs = 16
    num_ensemble_models = 1
    active_ratio = 1.0
    split_ratio = 0.5
    disparity_ratio = 0.25
    eps = 0.05
    augment = False
    refine_models = False
    train_disguised_model = False
    blind_strategy = False
    # not a flag of the reference: embed the noisy pair occurrences (2 P n_noise images per iteration, the bulk of its
    # work) in the feature model's 16-bit SCREENING mode and re-embed in the exact mode only the pairs whose side of a
    # cut is uncertain (settle.py); needs a feature model with `process_screen`.  OFF by default (round 6): the reference's
    # semantics are exact (every embedding in one arithmetic, code/ALINK_arc.py:154-167) and so is the default here;
    # --screen_settle buys ~2.2x on an iteration under a MEASURED and sample-audited error bound — the same query set and
    # fine-tune data on every workload measured, a statistical guarantee, not a theorem.  When it is on the
    # loop prints each iteration's audit.
    screen_settle = False
    # not a flag of the reference: keyword options for settle.select_queries_settled (safety, delta0, min_sample, audit, ...)
    settle_options = None

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(type(self), k):
                raise AttributeError("unknown flag %r" % k)
            setattr(self, k, v)


def add_flags(parser):
    """Register the reference's flags on an argparse parser (tf.flags is not a dependency here)."""
    for name in sorted(n for n in vars(Flags) if not n.startswith("_")):
        default = getattr(Flags, name)
        if default is None:                     # structured options: set on the Flags object, not on the command line
            continue
        if isinstance(default, bool):
            parser.add_argument("--" + name, action="store_true", default=default)
        else:
            parser.add_argument("--" + name, type=type(default), default=default)
    return parser


class LoopState(object):
    """What the drivers keep in module globals: ACTIVE_COUNT, UN_SIZE and the pending fine-tune set."""

    def __init__(self):
        self.active_count = 0
        self.un_size = 0
        self.left = np.array([])
        self.right = np.array([])
        self.y = np.array([])
        self.finetunes = 0
        self.iterations = 0
        self.history = []
        self.settle_info = []
        self.recalibrations = 0          # iterations during which the feature model re-calibrated itself (split precision)


def _np(x):
    """host ndarray of a NumPy array or a (CUDA) torch tensor"""
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _concat(old, parts):
    parts = [_np(p) for p in parts]
    return np.concatenate(([old] if np.asarray(old).shape[0] > 0 else []) + parts)


def _rows(a, idx):
    if hasattr(a, "detach"):
        import torch
        return a[torch.as_tensor(np.asarray(idx, np.int64), device=a.device)]
    return np.asarray(a)[np.asarray(idx, np.int64)]


def _set_rows(a, idx, v):
    if hasattr(a, "detach"):
        import torch
        a[torch.as_tensor(np.asarray(idx, np.int64), device=a.device)] = torch.as_tensor(v, device=a.device).to(a.dtype)
    else:
        a[np.asarray(idx, np.int64)] = _np(v)


class Recalibrated(RuntimeError):
    """raised by alink_iteration(group=...) on EVERY rank when some rank's feature model re-calibrated its split-precision
    scales during the iteration: the iteration's counters are rolled back; run_alink_dfw merges the scales
    (distributed.merge_calibration) and runs the iteration once more under them"""


def sync_host_randomness(ensembleNoise, shards):
    """Make rank 0's host-side random state every rank's: NumPy's global stream (the balanced generator's sampling,
    fit()'s shuffles and SmallRes' dropout masks draw from it, as the reference's do: SURVEY.md §5) and the stream
    state of every noise object.  After it, ranks that run the same host code stay in step by themselves."""
    states = [z.stream_state() if hasattr(z, "stream_state") else None for z in ensembleNoise]
    rng, states = shards.bcast((np.random.get_state(), states))
    np.random.set_state(rng)
    for z, st in zip(ensembleNoise, states):
        if st is not None and hasattr(z, "set_stream_state"):
            z.set_stream_state(st)


def alink_iteration(state, flags, batch_x, batch_y, batch_x_features, bag, ensembleNoise, student, dataGen,
                    noisy_for_student, clean_for_student, image_res, col=0, verbose=1, labels_one_hot=False,
                    noisy_for_student_screen=None, group=None, batch_x_rows=None, calibration_of=None):
    """One pass of the loop body (code/ALINK_arc.py:150-254) over an already-built mini-batch.

    batch_x            [left, right] pair images (P, H, W, 3)
    batch_y            (P, 1) oracle labels (1 = same identity)
    batch_x_features   [left, right] teacher features of the clean pairs
    noisy_for_student  f(noisy_images) -> what the student consumes (features for the DFW drivers,
                       the low-res pixels themselves for Multi-PIE)
    clean_for_student  [left, right] clean inputs of the student (features / low-res pixels)
    noisy_for_student_screen  the same conversion in the feature model's fast 16-bit mode: when given (and
                       flags.screen_settle), every noisy copy is converted by it first and only the pairs whose side of
                       a cut of the selection rule is uncertain — plus the pairs that end up selected — are converted
                       again by `noisy_for_student` (settle.select_queries_settled): query set, oracle count, labels and
                       the rows that reach the fine-tune set equal the all-exact run's on every workload measured, with a
                       sampled audit behind the error bound (settle.py).
    group              one process per GPU (not in the reference, which is one process on one GPU): a torch.distributed
                       ProcessGroup (torch.distributed.group.WORLD, or True, for all ranks).  The P pair rows — 2 P n_noise
                       noisy embeddings, the bulk of an iteration — are split contiguously over the ranks
                       (distributed.RowShards): every rank perturbs ITS rows (noise keyed by the global row: what a row
                       receives does not depend on the number of ranks), converts and scores them; ONE all-gather carries
                       the (P, 2) student predictions of all noises (8 B per row and noise — never the features); the
                       selection rule runs replicated on every rank (it is deterministic); a settle request is served by
                       the rank that owns the row; the rows the fine-tune set takes from the noisy passes are gathered
                       from their owners (a few hundred feature vectors); the fine-tune itself runs on every rank through
                       distributed.dp_train_on_batch (replicated at the reference's batch of 16, gradient all-reduce
                       above DP_SHARD_MIN_ROWS rows) with rank 0's shuffles.  Everything else (the clean pass: a few dozen
                       unique images; committee predictions; bookkeeping) is replicated.  Query list, oracle count,
                       fine-tune set and the student's weights afterwards equal the single-process iteration's bit for
                       bit (tests/test_alink_multirank.py over gloo at world sizes 2 and 3; tests/test_gpu_distributed.py
                       with two processes on one card).
    batch_x_rows       (lo, hi) when batch_x holds only THIS rank's rows lo : hi of the P pairs (run_alink_dfw gathers
                       just those: 150 KB of pixels per row and side); None: batch_x holds all P rows.
    calibration_of     optional f() -> a picklable calibration state of the feature model (ArcFace: the split-precision
                       scales): checked equal on all ranks before and after the iteration — a rank that re-calibrated on
                       its own rows would embed to different last bits than its peers.
    Returns the number of examples added to the pending fine-tune set, or -1 when the reference
    `continue`s (no query survived: code/ALINK_arc.py:203-205 — the stop check is skipped too).
    """
    log = print if verbose else (lambda *a, **k: None)
    P = len(batch_y)
    shards = None
    lo, hi = 0, P
    if group is not None:
        from . import distributed as _D
        group = _D.resolve_group(group)
        shards = _D.RowShards(P, group)
        lo, hi = shards.lo, shards.hi
        sync_host_randomness(ensembleNoise, shards)
        cal0 = calibration_of() if calibration_of is not None else None
        if calibration_of is not None and not shards.same_everywhere(cal0):
            raise RuntimeError("alink_iteration: the feature model's calibration state differs between ranks: calibrate on one rank "
                               "(or on a sample every rank draws identically) and distributed.broadcast_calibration() it")
    elif calibration_of is not None:
        cal0 = calibration_of()
    if batch_x_rows is not None:
        assert tuple(batch_x_rows) == (lo, hi), "batch_x_rows %s is not this rank's shard %s" % (tuple(batch_x_rows), (lo, hi))
        local_x = batch_x
    elif shards is not None:
        local_x = [batch_x[0][lo:hi], batch_x[1][lo:hi]]
    else:
        local_x = batch_x
    state.iterations += 1
    state.un_size += P
    ensemblePredictions = _np(bag.predict(batch_x_features))
    m1_labels = np.argmax(ensemblePredictions, axis=1)
    if labels_one_hot:                      # ALINK_MTP.py:174 passes keras.utils.to_categorical(..., 2)
        m1_labels = helpers.one_hot(m1_labels, 2)
    # With ranks, whatever THIS rank does between two exchanges may fail on this rank alone — a Poisson lam < 0 in its rows, a
    # 16-bit screening forward that leaves its range, an out-of-memory — while its peers are on their way into the next
    # all-gather.  Every such phase runs under `guard`: a failure is recorded with the shards (RowShards.fail), the phase
    # yields None, the rank feeds zeros of the right shape into the exchange, and EVERY rank raises there together
    # (until then only the settle requests were covered and the peers of a failed rank blocked in the collective).
    broken = []

    def guard(fn):
        if shards is None:
            return fn()
        if broken:
            return None
        try:
            return fn()
        except Exception as exc:
            shards.fail("%s: %s" % (type(exc).__name__, exc))
            broken.append(exc)
            return None

    if shards is not None:
        noisy_data = guard(lambda: bag.attackModel(local_x, image_res, m1_labels[lo:hi], rows=(lo, P)))
    else:
        noisy_data = bag.attackModel(local_x, image_res, m1_labels)
    n_noise = len(ensembleNoise)
    pred_shape = ensemblePredictions.shape[1:]

    def convert(fn, p):
        """what the student consumes for this rank's noisy rows; an empty shard converts nothing (the feature model is not
        called on a (0, H, W, 3) batch) and yields an empty array shaped like the clean inputs' rows"""
        if len(p) == 0:
            return np.zeros((0,) + tuple(_np(clean_for_student[0][:1]).shape[1:]), _np(clean_for_student[0][:1]).dtype)
        return fn(p)

    def predict_rows(sides):
        """the student's predictions for rows this rank holds (an empty shard predicts nothing)"""
        if len(sides[0]) == 0:
            return np.zeros((0,) + tuple(pred_shape), np.float32)
        return np.asarray(_np(student.predict(sides)), np.float32)

    def all_noises(local_preds):
        """[per noise (rows of this rank, C)] -> [per noise (P, C)]: one all-gather for all noises"""
        if shards is None:
            return local_preds
        full = shards.all_rows(np.stack(local_preds, axis=1))             # (P, n_noise, C)
        return [np.ascontiguousarray(full[:, jj]) for jj in range(n_noise)]

    if noisy_for_student_screen is not None and getattr(flags, "screen_settle", False):
        from . import settle
        pixels = noisy_data
        noisy_data = guard(lambda: [[convert(noisy_for_student_screen, p) for p in part] for part in pixels])
        noisy_data = guard(lambda: [[f.clone() if hasattr(f, "detach") else np.array(f, copy=True) for f in part] for part in noisy_data])
        local_preds = guard(lambda: [predict_rows([noisy_data[0][jj], noisy_data[1][jj]]) for jj in range(n_noise)])
        if local_preds is None:                # this rank failed above: zeros into the exchange, where every rank raises
            local_preds = [np.zeros((hi - lo,) + tuple(pred_shape), np.float32) for _ in range(n_noise)]
        screened = all_noises(local_preds)

        def settle_many(requests):
            """a round's requests [(noise, pairs)]: both sides of every request converted in ONE exact call — by the
            rank that owns the rows; the exact predictions of all requests then travel in one all-gather"""
            mine = [(jj, np.asarray(idx, np.int64) - lo if shards is None else shards.owned(idx)) for jj, idx in requests]

            def local_part():
                parts = [_rows(pixels[s][jj], own) for jj, own in mine for s in (0, 1) if len(own)]
                conv = None
                if parts:
                    if hasattr(parts[0], "detach"):
                        import torch
                        conv = noisy_for_student(torch.cat(parts))
                    else:
                        conv = noisy_for_student(np.concatenate([np.asarray(p) for p in parts]))
                out, o = [], 0
                for jj, own in mine:
                    sides = []
                    for s in (0, 1):
                        if len(own):
                            sides.append(conv[o:o + len(own)])
                            _set_rows(noisy_data[s][jj], own, sides[-1])     # the exact rows replace the screened ones
                            o += len(own)
                    out.append(predict_rows(sides) if len(own) else np.zeros((0,) + tuple(pred_shape), np.float32))
                return out
            if shards is None:
                return local_part()
            try:
                out = local_part()
            except Exception as exc:          # this rank's peers are on their way into the exchange: fail THERE, on every rank
                shards.fail("%s: %s" % (type(exc).__name__, exc))
                out = [np.zeros((len(own),) + tuple(pred_shape), np.float32) for _, own in mine]
            return shards.subsets([idx for _, idx in requests], out, pred_shape)
        queryIndices, active, labels, disguisedPredictions, _, info = settle.select_queries_settled(
            ensemblePredictions, screened, batch_y, None, col=col, disparity_ratio=flags.disparity_ratio,
            eps=flags.eps, blind_strategy=flags.blind_strategy, settle_many=settle_many,
            **dict(getattr(flags, "settle_options", None) or {}))
        if shards is not None:
            info = dict(info, rank=shards.rank, world=shards.world, rows_of_this_rank=hi - lo)
        state.settle_info.append(info)
        if verbose:
            aud = info.get("audit") or {}
            print("screen-then-settle: %.1f %% of the (pair, noise) rows re-embedded exactly, error bound %.2e; audit: %s sampled, "
                  "largest error seen %s, %s beyond the bound%s" % (100.0 * info.get("fraction_settled", 0.0), info.get("delta", float("nan")),
                                                                     aud.get("m", 0), aud.get("max_err", "n/a"), aud.get("exceedances", "n/a"),
                                                                     " (bound widened, pass repeated)" if info.get("widened") else ""))
    else:
        if shards is None:
            noisy_data = [[convert(noisy_for_student, p) for p in part] for part in noisy_data]
            local_preds = [predict_rows([noisy_data[0][jj], noisy_data[1][jj]]) for jj in range(n_noise)]
        else:
            pixels = noisy_data
            noisy_data = guard(lambda: [[convert(noisy_for_student, p) for p in part] for part in pixels])
            local_preds = guard(lambda: [predict_rows([noisy_data[0][jj], noisy_data[1][jj]]) for jj in range(n_noise)])
            if local_preds is None:
                local_preds = [np.zeros((hi - lo,) + tuple(pred_shape), np.float32) for _ in range(n_noise)]
        disguisedPredictions = all_noises(local_preds)
        queryIndices, active, labels = selection.select_queries(
            ensemblePredictions, disguisedPredictions, batch_y, col=col, disparity_ratio=flags.disparity_ratio,
            eps=flags.eps, blind_strategy=flags.blind_strategy)
    if shards is not None and calibration_of is not None:
        if not shards.all_true(calibration_of() == cal0):
            # nothing of this iteration has been committed yet beyond these three: rolled back, so that the caller can merge the
            # ranks' scales and run it again (run_alink_dfw does, once)
            state.iterations -= 1
            state.un_size -= P
            if noisy_for_student_screen is not None and getattr(flags, "screen_settle", False):
                state.settle_info.pop()
            raise Recalibrated("alink_iteration: a rank's feature model re-calibrated itself during the iteration (a batch left the "
                               "split-precision range): ranks no longer embed to the same bits.  Calibrate on noisy images like "
                               "these first, then distributed.broadcast_calibration()")
    elif calibration_of is not None and calibration_of() != cal0:
        # one process: the results stay the exact mode's (scales are powers of two: ~2e-7 on an embedding), but the clean pass
        # and the rows embedded after the change used different scales — counted, so that a caller can calibrate on noisier images
        state.recalibrations += 1
        log("note: the feature model re-calibrated its split-precision scales during this iteration")
    state.active_count += active
    log("Active Count so far : %d" % state.active_count)
    if len(queryIndices) == 0:
        return -1
    q = np.asarray(queryIndices)
    mp = int(len(q) / float(n_noise))
    chunks = [q[i * mp:(i + 1) * mp] for i in range(n_noise)]
    if shards is None:
        noisy_left = [_np(noisy_data[0][i])[chunks[i]] for i in range(n_noise)]
        noisy_right = [_np(noisy_data[1][i])[chunks[i]] for i in range(n_noise)]
    else:
        # chunk i of the query list takes noise i's rows (code/ALINK_arc.py:213-222): gathered from the ranks that own them
        row_shape = tuple(_np(clean_for_student[0][:1]).shape[1:])
        mine_rows = guard(lambda: [_np(_rows(noisy_data[s][i], shards.owned(chunks[i]))) for s in (0, 1) for i in range(n_noise)])
        if mine_rows is None:
            mine_rows = [np.zeros((len(shards.owned(chunks[i])),) + tuple(row_shape), _np(clean_for_student[0][:1]).dtype)
                         for s in (0, 1) for i in range(n_noise)]
        got = shards.subsets(chunks + chunks, mine_rows, row_shape, _np(clean_for_student[0][:1]).dtype)
        noisy_left, noisy_right = got[:n_noise], got[n_noise:]
    state.left = _concat(state.left, noisy_left)
    state.right = _concat(state.right, noisy_right)
    state.y = _concat(state.y, [labels[i * mp:(i + 1) * mp] for i in range(n_noise)])
    added = n_noise * mp
    if state.y.shape[0] >= flags.batch_send:
        (X_old_left, X_old_right), Y_old = next(dataGen)
        for _ in range(flags.mixture_ratio - 1):
            X_old_temp, Y_old_temp = next(dataGen)
            X_old_left = np.concatenate((X_old_left, X_old_temp[0]))
            X_old_right = np.concatenate((X_old_right, X_old_temp[1]))
            Y_old = np.concatenate((Y_old, Y_old_temp))
        if flags.augment:
            raise NotImplementedError("--augment (tf.contrib rotations + imgaug, code/helpers.py:114-141) is outside "
                                      "the hot path and not built")
        left = np.concatenate((state.left, _np(clean_for_student[0])[q], X_old_left))
        right = np.concatenate((state.right, _np(clean_for_student[1])[q], X_old_right))
        y = np.concatenate((state.y, labels, Y_old))
        net = getattr(student, "siamese_net", None)
        if shards is not None and net is not None and hasattr(net, "dp_group"):
            net.dp_group = group              # fit() -> distributed.dp_train_on_batch, rank 0's shuffles
        try:
            hist = student.finetune([left, right], y, flags.ft_epochs, 16, 1 if verbose else 0)
        finally:
            if shards is not None and net is not None and hasattr(net, "dp_group"):
                net.dp_group = None
        state.history.append(hist)
        state.finetunes += 1
        state.left, state.right, state.y = np.array([]), np.array([]), np.array([])
    return added


def _embed_pairs_unique(conversionModel, plain_part, disguise_part, on_device):
    """Teacher features of createMiniBatch(plain_part, disguise_part) with every image embedded once.
    on_device: keep pixels, pair gathers and features as CUDA tensors, so that the P pair occurrences
    (hundreds of MB of pixels per side) never cross PCIe — noise, resize and embedding all take tensors.
    Replicated on every rank of a multi-rank loop (a few dozen unique images: 80 of an iteration's 30,800 embeddings)."""
    n_plain = [len(p) for p in plain_part]
    n_dig = [len(d) for d in disguise_part]
    li, ri, y = pairs.createMiniBatchIndices(n_plain, n_dig)
    unique = np.concatenate([np.asarray(p) for p in plain_part] + [np.asarray(d) for d in disguise_part])
    if on_device:
        import torch
        unique = torch.from_numpy(np.ascontiguousarray(unique, dtype=np.float32)).cuda()
        li, ri = torch.from_numpy(li).long().cuda(), torch.from_numpy(ri).long().cuda()
    feats = conversionModel.process(unique)
    return unique, li, ri, y, feats


def _calibration_probe(conversionModel):
    """f() -> the feature model's calibration state (split-precision scales), or None when it has none to keep in step"""
    bb = getattr(getattr(conversionModel, "model", None), "model", None)
    if bb is not None and hasattr(bb, "state") and getattr(bb, "dtype", None) == "f16x2":
        return bb.state
    return None


def run_alink_dfw(flags, conversionModel, bag, ensembleNoise, disguisedFacesModel, X_plain_raw, X_dig_post, dataGen,
                  image_res, col=0, verbose=1, state=None, on_device=True, group=None):
    """The framework loop of ALINK_arc.py (col = 0) / ALINK.py (col = 1): code/ALINK_arc.py:139-260.
    X_plain_raw / X_dig_post: per-person lists of raw images (k_i, H, W, 3).  Returns LoopState.
    group (one process per GPU; see alink_iteration): every rank is handed the SAME data and models (same weights, same
    calibration) and calls this together; each materialises only its own rows of an iteration's pair batch, the state
    returned — and the student's weights — are the same on every rank and equal the single-process loop's.  Rank 0
    alone writes `flags.out_model`."""
    log = print if verbose else (lambda *a, **k: None)
    assert 0 <= flags.disparity_ratio <= 1 and 0 <= flags.eps < 0.5
    state = state or LoopState()
    rank = 0
    if group is not None:
        from . import distributed as _D
        group = _D.resolve_group(group)
        rank = _D._dist().get_rank(group)
    log("== Framework beginning with a pool of %d" % (len(X_dig_post)))
    for ii in range(0, len(X_dig_post), flags.alink_bs):
        log("\nIteration #%d" % ((ii // flags.alink_bs) + 1))
        plain_part = X_plain_raw[ii: ii + flags.alink_bs]
        disguise_part = X_dig_post[ii: ii + flags.alink_bs]
        unique, li, ri, batch_y, feats = _embed_pairs_unique(conversionModel, plain_part, disguise_part, on_device)
        rows = None
        if group is not None:                   # only this rank's rows of the pair batch are gathered (150 KB per row and side)
            rows = _D.shard_range(len(batch_y), rank, _D._dist().get_world_size(group))
            batch_x = [unique[li[rows[0]:rows[1]]], unique[ri[rows[0]:rows[1]]]]
        else:
            batch_x = [unique[li], unique[ri]]
        batch_x_features = [feats[li], feats[ri]]
        probe = _calibration_probe(conversionModel)
        streams = [z.stream_state() if hasattr(z, "stream_state") else None for z in ensembleNoise]
        host_rng = np.random.get_state()
        for attempt in (0, 1):
            try:
                added = alink_iteration(state, flags, batch_x, batch_y, batch_x_features, bag, ensembleNoise,
                                        disguisedFacesModel, dataGen, noisy_for_student=conversionModel.process,
                                        clean_for_student=batch_x_features, image_res=image_res, col=col, verbose=verbose,
                                        noisy_for_student_screen=getattr(conversionModel, "process_screen", None),
                                        group=group, batch_x_rows=rows, calibration_of=probe)
                break
            except Recalibrated:
                #  noisy images left the calibrated range on some rank, which lowered ITS scales: every rank takes
                # the elementwise minimum (distributed.merge_calibration, what committee_pool_topk_settled does), the
                # iteration's random streams are wound back, the clean pass is embedded again under the merged scales and
                # the iteration runs once more; a second change raises
                if attempt or group is None:
                    raise
                _D.merge_calibration([conversionModel.model.model], group=group)
                state.recalibrations += 1
                log("note: a rank re-calibrated its split-precision scales; scales merged over the ranks, iteration repeated")
                np.random.set_state(host_rng)
                for z, st in zip(ensembleNoise, streams):
                    if st is not None and hasattr(z, "set_stream_state"):
                        z.set_stream_state(st)
                unique, li, ri, batch_y, feats = _embed_pairs_unique(conversionModel, plain_part, disguise_part, on_device)
                batch_x = [unique[li[rows[0]:rows[1]]], unique[ri[rows[0]:rows[1]]]]
                batch_x_features = [feats[li], feats[ri]]
        if added < 0:
            continue
        if int(flags.active_ratio * state.un_size) <= state.active_count:
            log("Specified limit reached! Stopping algorithm")
            break
    log("Active Count: %d out of %d" % (state.active_count, state.un_size))
    if flags.out_model and rank == 0:
        disguisedFacesModel.save(flags.out_model)
    return state


def createMiniBatchMTP(X_dig):
    """readMTP.createMiniBatch (code/readMTP.py:123-135): all (person i x person j) image pairs."""
    X_left, X_right, Y = [], [], []
    for i in range(len(X_dig)):
        for j in range(len(X_dig)):
            for x in X_dig[i]:
                for y in X_dig[j]:
                    X_left.append(x)
                    X_right.append(y)
                    Y.append([1] if i == j else [0])
    return [np.stack(X_left), np.stack(X_right)], np.stack(Y)


def run_alink_mtp(flags, conversionModel, bag, ensembleNoise, lowResModel, X_dig_post, dataGen, image_res, low_res,
                  verbose=1, state=None, group=None):
    """The framework loop of ALINK_MTP.py (code/ALINK_MTP.py:150-266): the teacher committee scores
    high-res features, the student (SmallRes) sees noisy LOW-res pixels and is fine-tuned on them.
    group: as in run_alink_dfw; the student's fine-tune goes through distributed.dp_train_on_batch like the head's (SmallResNet
    has the same four-method side of it): replicated at the reference's batch of 16 (every rank holds the same SmallRes and draws
    the same mask seed), sharded with one exchange of the 20 MB gradient buffer from SmallResNet.DP_SHARD_MIN_ROWS rows."""
    from . import noise as _noise
    log = print if verbose else (lambda *a, **k: None)
    state = state or LoopState()
    rank = 0
    if group is not None:
        from . import distributed as _D
        group = _D.resolve_group(group)
        rank = _D._dist().get_rank(group)
    log("== Framework beginning with a pool of %d ==" % (len(X_dig_post)))
    for ii in range(0, len(X_dig_post), flags.alink_bs):
        log("\nIteration #%d" % ((ii // flags.alink_bs) + 1))
        part = X_dig_post[ii: ii + flags.alink_bs]
        n = [len(p) for p in part]
        off = np.concatenate([[0], np.cumsum(n)])
        unique = np.concatenate([np.asarray(p) for p in part])
        li = np.concatenate([np.repeat(np.arange(off[i], off[i + 1]), n[j]) for i in range(len(n)) for j in range(len(n))])
        ri = np.concatenate([np.tile(np.arange(off[j], off[j + 1]), n[i]) for i in range(len(n)) for j in range(len(n))])
        batch_y = np.concatenate([np.full(n[i] * n[j], 1 if i == j else 0) for i in range(len(n))
                                  for j in range(len(n))]).reshape(-1, 1)
        high = np.asarray(_noise.resize_images(unique, image_res))           # readMTP.resizeImages (:164-165)
        low = np.asarray(_noise.resize_images(unique, low_res))
        feats = np.asarray(conversionModel.process(high))
        rows = None
        if group is not None:
            rows = _D.shard_range(len(batch_y), rank, _D._dist().get_world_size(group))
            batch_x = [unique[li[rows[0]:rows[1]]], unique[ri[rows[0]:rows[1]]]]
        else:
            batch_x = [unique[li], unique[ri]]
        added = alink_iteration(state, flags, batch_x, batch_y, [feats[li], feats[ri]], bag, ensembleNoise, lowResModel,
                                dataGen, noisy_for_student=lambda p: np.asarray(p), clean_for_student=[low[li], low[ri]],
                                image_res=low_res, col=0, verbose=verbose, labels_one_hot=True, group=group, batch_x_rows=rows)
        if added < 0:
            log("== Nothing in this set. Skipping batch ==")
            continue
        if int(flags.active_ratio * state.un_size) <= state.active_count:
            log("== Specified limit reached! Stopping algorithm ==")
            break
    log("== Active Count: %d out of %d ==" % (state.active_count, state.un_size))
    if flags.out_model and rank == 0:
        lowResModel.save(flags.out_model)
    return state


def top1_identification(lowResModel, X_test, chunk_pairs=8192):
    """code/ALINK_MTP.py:274-289, including its argmax over the squeezed (G, 2) score array (the
    flattened index is compared with the person id, as the reference does).  The reference scores one probe image against
    the gallery per predict() call; here `chunk_pairs` (probe, gallery) pairs go into one call — the same pairs, the same
    argmax per probe."""
    X_gallery = [x[0] for x in X_test]
    gal = np.array(X_gallery)
    G = len(X_gallery)
    probes = [(i, np.asarray(x)) for i in range(len(X_test)) for x in X_test[i]]
    per_call = max(1, int(chunk_pairs) // max(G, 1))
    acc = 0
    for s0 in range(0, len(probes), per_call):
        part = probes[s0:s0 + per_call]
        left = np.repeat(np.stack([x for _, x in part]), G, axis=0)
        right = np.concatenate([gal] * len(part), axis=0)
        scores = np.asarray(lowResModel.predict([left, right])).reshape(len(part), -1)       # (probe, G * C): the squeezed array, flattened
        predicted = np.argmax(scores, axis=1)
        acc += int(sum(int(p) == i for p, (i, _) in zip(predicted, part)))
    return acc / float(len(probes))


def pretrain(model, dataGen, epochs, batch_size, n_steps=320000, refine=False, verbose=1):
    """The pre-training branches of the drivers (code/ALINK_arc.py:96-137): load if saved, else (or
    when refining) customTrainModel on the balanced generator and save.  Returns True if trained."""
    loaded = model.maybeLoadFromMemory()
    if loaded and not refine:
        return False
    model.customTrainModel(dataGen, epochs, batch_size, 0.2, n_steps=n_steps, verbose=verbose)
    model.save()
    return True

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
    # cut is uncertain (settle.py) — same query set, same fine-tune data; needs a feature model with `process_screen`
    screen_settle = True

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(type(self), k):
                raise AttributeError("unknown flag %r" % k)
            setattr(self, k, v)


def add_flags(parser):
    """Register the reference's flags on an argparse parser (tf.flags is not a dependency here)."""
    for name in sorted(n for n in vars(Flags) if not n.startswith("_")):
        default = getattr(Flags, name)
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


def alink_iteration(state, flags, batch_x, batch_y, batch_x_features, bag, ensembleNoise, student, dataGen,
                    noisy_for_student, clean_for_student, image_res, col=0, verbose=1, labels_one_hot=False,
                    noisy_for_student_screen=None):
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
                       the rows that reach the fine-tune set equal the all-exact run's.
    Returns the number of examples added to the pending fine-tune set, or -1 when the reference
    `continue`s (no query survived: code/ALINK_arc.py:203-205 — the stop check is skipped too).
    """
    log = print if verbose else (lambda *a, **k: None)
    state.iterations += 1
    state.un_size += len(batch_x[0])
    ensemblePredictions = _np(bag.predict(batch_x_features))
    m1_labels = np.argmax(ensemblePredictions, axis=1)
    if labels_one_hot:                      # ALINK_MTP.py:174 passes keras.utils.to_categorical(..., 2)
        m1_labels = helpers.one_hot(m1_labels, 2)
    noisy_data = bag.attackModel(batch_x, image_res, m1_labels)
    n_noise = len(ensembleNoise)
    if noisy_for_student_screen is not None and getattr(flags, "screen_settle", True):
        from . import settle
        pixels = noisy_data
        noisy_data = [[noisy_for_student_screen(p) for p in part] for part in pixels]
        noisy_data = [[f.clone() if hasattr(f, "detach") else np.array(f, copy=True) for f in part] for part in noisy_data]
        screened = [_np(student.predict([noisy_data[0][jj], noisy_data[1][jj]])) for jj in range(n_noise)]

        def settle_many(requests):
            """a round's requests [(noise, pairs)]: both sides of every request converted in ONE exact call"""
            parts = [_rows(pixels[s][jj], idx) for jj, idx in requests for s in (0, 1)]
            if hasattr(parts[0], "detach"):
                import torch
                conv = noisy_for_student(torch.cat(parts))
            else:
                conv = noisy_for_student(np.concatenate([np.asarray(p) for p in parts]))
            out, o = [], 0
            for jj, idx in requests:
                sides = []
                for s in (0, 1):
                    sides.append(conv[o:o + len(idx)])
                    _set_rows(noisy_data[s][jj], idx, sides[-1])     # the exact rows replace the screened ones
                    o += len(idx)
                out.append(_np(student.predict(sides)))
            return out
        queryIndices, active, labels, disguisedPredictions, _, info = settle.select_queries_settled(
            ensemblePredictions, screened, batch_y, None, col=col, disparity_ratio=flags.disparity_ratio,
            eps=flags.eps, blind_strategy=flags.blind_strategy, settle_many=settle_many)
        state.settle_info.append(info)
    else:
        noisy_data = [[noisy_for_student(p) for p in part] for part in noisy_data]
        disguisedPredictions = [_np(student.predict([noisy_data[0][jj], noisy_data[1][jj]])) for jj in range(n_noise)]
        queryIndices, active, labels = selection.select_queries(
            ensemblePredictions, disguisedPredictions, batch_y, col=col, disparity_ratio=flags.disparity_ratio,
            eps=flags.eps, blind_strategy=flags.blind_strategy)
    state.active_count += active
    log("Active Count so far : %d" % state.active_count)
    if len(queryIndices) == 0:
        return -1
    q = np.asarray(queryIndices)
    mp = int(len(q) / float(n_noise))
    state.left = _concat(state.left, [_np(noisy_data[0][i])[q[i * mp:(i + 1) * mp]] for i in range(n_noise)])
    state.right = _concat(state.right, [_np(noisy_data[1][i])[q[i * mp:(i + 1) * mp]] for i in range(n_noise)])
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
        hist = student.finetune([left, right], y, flags.ft_epochs, 16, 1 if verbose else 0)
        state.history.append(hist)
        state.finetunes += 1
        state.left, state.right, state.y = np.array([]), np.array([]), np.array([])
    return added


def _embed_pairs_unique(conversionModel, plain_part, disguise_part, on_device):
    """Teacher features of createMiniBatch(plain_part, disguise_part) with every image embedded once.
    on_device: keep pixels, pair gathers and features as CUDA tensors, so that the P pair occurrences
    (hundreds of MB of pixels per side) never cross PCIe — noise, resize and embedding all take tensors."""
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


def run_alink_dfw(flags, conversionModel, bag, ensembleNoise, disguisedFacesModel, X_plain_raw, X_dig_post, dataGen,
                  image_res, col=0, verbose=1, state=None, on_device=True):
    """The framework loop of ALINK_arc.py (col = 0) / ALINK.py (col = 1): code/ALINK_arc.py:139-260.
    X_plain_raw / X_dig_post: per-person lists of raw images (k_i, H, W, 3).  Returns LoopState."""
    log = print if verbose else (lambda *a, **k: None)
    assert 0 <= flags.disparity_ratio <= 1 and 0 <= flags.eps < 0.5
    state = state or LoopState()
    log("== Framework beginning with a pool of %d" % (len(X_dig_post)))
    for ii in range(0, len(X_dig_post), flags.alink_bs):
        log("\nIteration #%d" % ((ii // flags.alink_bs) + 1))
        plain_part = X_plain_raw[ii: ii + flags.alink_bs]
        disguise_part = X_dig_post[ii: ii + flags.alink_bs]
        unique, li, ri, batch_y, feats = _embed_pairs_unique(conversionModel, plain_part, disguise_part, on_device)
        batch_x = [unique[li], unique[ri]]
        batch_x_features = [feats[li], feats[ri]]
        added = alink_iteration(state, flags, batch_x, batch_y, batch_x_features, bag, ensembleNoise,
                                disguisedFacesModel, dataGen, noisy_for_student=conversionModel.process,
                                clean_for_student=batch_x_features, image_res=image_res, col=col, verbose=verbose,
                                noisy_for_student_screen=getattr(conversionModel, "process_screen", None))
        if added < 0:
            continue
        if int(flags.active_ratio * state.un_size) <= state.active_count:
            log("Specified limit reached! Stopping algorithm")
            break
    log("Active Count: %d out of %d" % (state.active_count, state.un_size))
    if flags.out_model:
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
                  verbose=1, state=None):
    """The framework loop of ALINK_MTP.py (code/ALINK_MTP.py:150-266): the teacher committee scores
    high-res features, the student (SmallRes) sees noisy LOW-res pixels and is fine-tuned on them."""
    from . import noise as _noise
    log = print if verbose else (lambda *a, **k: None)
    state = state or LoopState()
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
        batch_x = [unique[li], unique[ri]]
        added = alink_iteration(state, flags, batch_x, batch_y, [feats[li], feats[ri]], bag, ensembleNoise, lowResModel,
                                dataGen, noisy_for_student=lambda p: np.asarray(p), clean_for_student=[low[li], low[ri]],
                                image_res=low_res, col=0, verbose=verbose, labels_one_hot=True)
        if added < 0:
            log("== Nothing in this set. Skipping batch ==")
            continue
        if int(flags.active_ratio * state.un_size) <= state.active_count:
            log("== Specified limit reached! Stopping algorithm ==")
            break
    log("== Active Count: %d out of %d ==" % (state.active_count, state.un_size))
    if flags.out_model:
        lowResModel.save(flags.out_model)
    return state


def top1_identification(lowResModel, X_test):
    """code/ALINK_MTP.py:274-289, including its argmax over the squeezed (G, 2) score array (the
    flattened index is compared with the person id, as the reference does)."""
    X_gallery = [x[0] for x in X_test]
    gal = np.array(X_gallery)
    total_count, acc = 0, 0
    for i in range(len(X_test)):
        for x in X_test[i]:
            left = np.repeat(np.asarray(x)[None], len(X_gallery), axis=0)
            predicted_scores = np.squeeze(lowResModel.predict([left, gal]))
            predicted_id = np.argmax(predicted_scores)
            total_count += 1
            if predicted_id == i:
                acc += 1
    return acc / float(total_count)


def pretrain(model, dataGen, epochs, batch_size, n_steps=320000, refine=False, verbose=1):
    """The pre-training branches of the drivers (code/ALINK_arc.py:96-137): load if saved, else (or
    when refining) customTrainModel on the balanced generator and save.  Returns True if trained."""
    loaded = model.maybeLoadFromMemory()
    if loaded and not refine:
        return False
    model.customTrainModel(dataGen, epochs, batch_size, 0.2, n_steps=n_steps, verbose=verbose)
    model.save()
    return True

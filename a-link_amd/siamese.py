"""siamese — drop-in for the hot-path classes of reference code/siamese.py.

    SiameseNetwork(shape, modelName, learningRate=1.0)      code/siamese.py:19-131
    ArcFace(shape, model_path)                              code/siamese.py:219-234
    RESNET50(shape)                                         code/siamese.py:203-216  (see resnet50.py)
    FaceVGG16(shape)                                        code/siamese.py:187-200  (see vgg16.py)
    SmallRes(imageShape, featureShape, name, learningRate)  code/siamese.py:134-184  (see smallres.py)

Same constructor signatures, attributes (`siamese_net`, `modelName`, `shape`, `learningRate`) and
methods.  `siamese_net` is a DenseHead (GPU) exposing the Keras Model calls the reference makes.
"""
import sys

import numpy as np

from . import face_model
from .head import DenseHead, EarlyStopping, ReduceLROnPlateau, to_categorical


class _Args(dict):
    """easydict.EasyDict stand-in for the 4 keys the reference passes (code/siamese.py:221-226)."""
    __getattr__ = dict.__getitem__


def _inverse_frequency_weights(y):
    """code/siamese.py:95-98: weight of class c = (n // count_c), the two normalised to sum 1.  The
    reference file has no `from __future__ import division`, so `len(y) / np.sum(y == c)` FLOORS under
    the Python 2 it was written for; an absent class divides by zero -> inf/nan weights there too."""
    y = np.asarray(y)
    with np.errstate(divide='ignore', invalid='ignore'):
        per_class = [np.int64(len(y)) // np.sum(y == c) for c in (0, 1)]
        total = float(per_class[0] + per_class[1])
        return {c: per_class[c] / total for c in (0, 1)}


class SiameseNetwork:
    _identity_preprocess = True
    _defer_metrics = True          # customTrainModel(verbose=0): steps enqueued, metrics read back in blocks (A/B switch for the test)
    _index_steps = True            # ... and, over this package's own generators, batches shipped as row indices (A/B switch)

    # what customTrainModel / finetune feed the head: this class one-hot labels with inverse-frequency class
    # weights (code/siamese.py:56,95-103); the baseline scorer of siamese3.py overrides both
    @staticmethod
    def _targets(y):
        return to_categorical(y, num_classes=2)

    @staticmethod
    def _step_class_weight(y):
        return _inverse_frequency_weights(y)

    def __init__(self, shape, modelName, learningRate=1.0, seed=None, adadelta_epsilon=1e-8, compute_dtype="f32"):
        self.learningRate = learningRate
        self.modelName = modelName
        self.shape = shape
        # abs(l - r) -> Dense(512, relu) -> Dense(64, relu) -> Dense(2) -> softmax,
        # loss binary_crossentropy, optimizer Adadelta(learningRate) (code/siamese.py:24-35)
        # compute_dtype="bf16": mixed-precision fine-tune (BASELINE configs[4]); the reference's Keras model is float32
        self.siamese_net = DenseHead(shape[0], 512, 64, lr=learningRate, rho=0.95, eps=adadelta_epsilon, seed=seed,
                                     compute_dtype=compute_dtype)

    def getDenseBarebones(self):
        """(code/siamese.py:37-42) layer specs as (units, activation) — there is no Keras here."""
        return [(512, 'relu'), (64, 'relu'), (2, None)]

    def finetune(self, X, Y, epochs, batch_size, verbose=1):
        early_stop = EarlyStopping(monitor='val_loss', min_delta=0.1, patience=5, verbose=1)
        reduce_lr = ReduceLROnPlateau(monitor='val_loss', factor=0.2, patience=5, min_lr=0.01, verbose=verbose)
        Y_encoded = self._targets(Y)
        return self.siamese_net.fit(self.preprocess(X), Y_encoded, batch_size=batch_size, epochs=epochs,
                                    validation_split=0.2, verbose=verbose, callbacks=[early_stop, reduce_lr])

    def testAccuracy(self, X, Y, batch_size=512):
        """(code/siamese.py:60-79) all-pairs accuracy; pairs gathered on device instead of stacked."""
        X = np.asarray(X, dtype=np.float32)
        Y = np.asarray(Y).ravel()
        n = len(X)
        li = np.repeat(np.arange(n, dtype=np.int32), n)
        ri = np.tile(np.arange(n, dtype=np.int32), n)
        probs = self.siamese_net.predict_device(X, X, li, ri).cpu().numpy()
        pred = np.argmax(probs, axis=1)
        truth = 1 * (Y[li] == Y[ri])
        return np.sum(pred == truth) / float(len(li))

    def customTrainModel(self, dataGen, epochs, batch_size, valRatio=0.2, n_steps=320000, preprocess=False,
                         verbose=1):
        """code/siamese.py:81-112: int(n_steps / batch_size) generator batches per epoch; each batch is
        permuted, its first int(n * valRatio) rows are held out for test_on_batch, the rest go through
        train_on_batch with inverse-frequency class weights.  Returns per-epoch
        (train loss, train acc, val loss, val acc) — each a sum over steps / steps_per_epoch, so a
        step without held-out rows counts as 0 in the val means, as in the reference."""
        steps_per_epoch = int(n_steps / batch_size)
        net = self.siamese_net
        logs = []
        # verbose = 0 on a DenseHead: nobody looks at a step's numbers before the epoch ends, so the steps are only ENQUEUED
        # (their {loss, accuracy} land in a device buffer read back every `block` steps and summed in step order: the same
        # sums) instead of synchronised one by one — 20,000 steps an epoch at the reference's settings
        deferred = (not verbose) and isinstance(net, DenseHead) and self._defer_metrics
        block = 256
        if deferred and self._index_steps and self._indexable(dataGen, preprocess):
            return self._custom_train_indexed(dataGen, epochs, steps_per_epoch, valRatio, block)
        for epoch in range(epochs):
            sums = np.zeros(4)                                   # tr loss, tr acc, vl loss, vl acc
            M, k = (net.torch.zeros((block, 4), dtype=net.torch.float32, device=net.device), 0) if deferred else (None, 0)

            def flush():
                nonlocal k
                if deferred and k:
                    for row in M[:k].cpu().numpy().astype(np.float64):
                        sums[:2] += row[:2]
                        sums[2:] += row[2:]
                    M.zero_()
                    k = 0
            for step in range(1, steps_per_epoch + 1):
                x, y = next(dataGen)
                if preprocess:
                    x = self.preprocess(x)
                order = np.random.permutation(len(y))
                n_held = int(len(y) * valRatio)
                held, used = order[:n_held], order[n_held:]
                if deferred:
                    net.train_on_batch([side[used] for side in x], self._targets(y[used]),
                                       class_weight=self._step_class_weight(y[used]), metrics_out=M[k, :2])
                    if n_held > 0:
                        net.test_on_batch([side[held] for side in x], self._targets(y[held]), metrics_out=M[k, 2:])
                    k += 1
                    if k == block:
                        flush()
                    continue
                sums[:2] += net.train_on_batch([side[used] for side in x], self._targets(y[used]),
                                               class_weight=self._step_class_weight(y[used]))[:2]
                if n_held > 0:
                    sums[2:] += net.test_on_batch([side[held] for side in x], self._targets(y[held]))[:2]
                if verbose:
                    sys.stdout.write("Epoch %d : %d / %d : Tr loss: %.4f, Tr acc: %.4f, Vl loss: %.4f, Vl acc: %.4f  \r"
                                     % ((epoch + 1, step, steps_per_epoch) + tuple(sums / step)))
                    sys.stdout.flush()
            flush()
            if verbose:
                print("\n")
            logs.append(tuple(sums / steps_per_epoch))
        return logs

    # -- customTrainModel over this package's own generators: the feature table stays on the device --------------------
    def _indexable(self, dataGen, preprocess):
        from .pairs import BalancedMix
        net = self.siamese_net
        if not (isinstance(dataGen, BalancedMix) and dataGen.indexable and net.dp_group is None):
            return False
        if preprocess and type(self).preprocess is not SiameseNetwork.preprocess:
            return False
        if type(net).train_on_batch is not DenseHead.train_on_batch or type(net).test_on_batch is not DenseHead.test_on_batch:
            return False                                       # an instrumented / overridden step must see every call
        t = dataGen.table()
        return t.ndim == 2 and t.shape[1] == net.d_in and len(t) < 2 ** 31

    def _custom_train_indexed(self, dataGen, epochs, steps_per_epoch, valRatio, block):
        """The loop of customTrainModel (code/siamese.py:91-110) with every feature resident on the device: the generator's
        table goes up once, a step is planned on the host as INDEX arithmetic only (the generator's balanced batch as rows of
        the table, np.random.permutation, the hold-out split, the inverse-frequency class weights — the same calls on the
        same random stream as the step-by-step form), `block` steps travel as one int32 + one float32 buffer and are
        enqueued by ONE call (alink_head_custom_train_steps); the metrics of a block are read back while the next one runs.
        Same kernels on the same rows in the same order: logs and weights equal the step-by-step form's bit for bit
        (tests/test_gpu_head.py)."""
        import ctypes as C
        from . import _abi
        net = self.siamese_net
        torch = net.torch
        tab = dataGen.table()
        cached = getattr(dataGen, "_device_table", None)
        if cached is None or cached[0] != net.device:
            td = tab.to(net.device, torch.float32).contiguous() if hasattr(tab, "detach") else \
                torch.from_numpy(np.ascontiguousarray(tab, dtype=np.float32)).to(net.device)
            cached = dataGen._device_table = (net.device, td)
        td = cached[1]
        od = net.out_dim
        base_targets = type(self)._targets is SiameseNetwork._targets
        base_weights = type(self)._step_class_weight is SiameseNetwork._step_class_weight
        cap_i, cap_f = block * 2 * 256, block * (od + 1) * 256
        sets = []
        for _ in range(2):
            sets.append({"i_host": torch.empty(cap_i, dtype=torch.int32).pin_memory(), "f_host": torch.empty(cap_f, dtype=torch.float32).pin_memory(),
                         "i_dev": torch.empty(cap_i, dtype=torch.int32, device=net.device), "f_dev": torch.empty(cap_f, dtype=torch.float32, device=net.device),
                         "M": torch.zeros((block, 4), dtype=torch.float32, device=net.device), "steps": 0, "up": torch.cuda.Event()})
        for st in sets:
            st["i_np"], st["f_np"] = st["i_host"].numpy(), st["f_host"].numpy()
        stream = torch.cuda.current_stream(net._tdev)

        # The launches of a block (5 per step) cost the host about as much as planning the block does: they run on a worker
        # thread (ctypes drops the GIL for the call) while this thread plans the next block.  One worker, jobs in order: the
        # stream sees the blocks in step order; a set's metrics are read only after its job has been handed over.
        import queue
        import threading
        jobs = queue.Queue()
        failure = []

        def launcher():
            while True:
                job = jobs.get()
                if job is None:
                    return
                st, d, k, weights = job
                try:
                    _abi.check(net.lib.alink_head_custom_train_steps(net.h, td.data_ptr(), st["i_dev"].data_ptr(), st["f_dev"].data_ptr(),
                                                                     d.ctypes.data_as(C.c_void_p), k, 1 if weights else 0,
                                                                     st["M"].data_ptr(), stream.cuda_stream), "alink_head_custom_train_steps")
                except Exception as e:                           # surfaces in the main thread at the next collect()
                    failure.append(e)
                finally:
                    st["launched"].set()
        worker = threading.Thread(target=launcher, daemon=True)
        worker.start()
        for st in sets:
            st["launched"] = threading.Event()
            st["launched"].set()

        def collect(st, sums):
            st["launched"].wait()
            if failure:
                raise failure[0]
            if st["steps"]:
                for row in st["M"][:st["steps"]].cpu().numpy().astype(np.float64):
                    sums[:2] += row[:2]
                    sums[2:] += row[2:]
                st["steps"] = 0

        def run(st, desc, k, io, fo, weights):
            """enqueue the k steps planned into set `st`"""
            if not k:
                return
            st["i_dev"][:io].copy_(st["i_host"][:io], non_blocking=True)
            st["f_dev"][:fo].copy_(st["f_host"][:fo], non_blocking=True)
            st["up"].record(stream)
            st["M"].zero_()
            st["launched"].clear()
            jobs.put((st, np.ascontiguousarray(desc[:k], dtype=np.int64), k, weights))
            st["steps"] = k

        try:
            return self._indexed_epochs(dataGen, epochs, steps_per_epoch, valRatio, block, sets, collect, run, od, cap_i, cap_f,
                                        base_targets, base_weights)
        finally:
            jobs.put(None)
            worker.join()
            for st in sets:
                st["launched"].wait()

    def _indexed_epochs(self, dataGen, epochs, steps_per_epoch, valRatio, block, sets, collect, run, od, cap_i, cap_f,
                        base_targets, base_weights):
        logs = []
        turn = 0
        for epoch in range(epochs):
            sums = np.zeros(4)
            done = 0
            while done < steps_per_epoch:
                st = sets[turn % 2]
                turn += 1
                collect(st, sums)                               # the block this set carried two turns ago (in step order)
                st["up"].synchronize()                          # its upload has left the pinned buffers
                nb = min(block, steps_per_epoch - done)
                inp, fnp = st["i_np"], st["f_np"]
                desc = np.empty((nb, 4), np.int64)
                io = fo = k = 0
                weights = None
                ended = None
                while k < nb:
                    try:
                        li, ri, y = dataGen.next_indices()
                    except StopIteration as e:                  # a finite source ended: the steps drawn so far still run
                        ended = e
                        break
                    n = len(y)
                    order = np.random.permutation(n)
                    n_held = int(n * valRatio)
                    yo = y[order]
                    flat = yo[n_held:, 0]
                    if base_weights:
                        c1 = int(np.count_nonzero(flat == 1))
                        c0 = int(np.count_nonzero(flat == 0))
                        if c0 and c1:
                            p0, p1 = (n - n_held) // c0, (n - n_held) // c1
                            tot = float(p0 + p1)
                            sw = np.where(flat == 1, p1 / tot, p0 / tot).astype(np.float32)
                        else:                                   # an absent class: the reference's inf / nan weights, the slow way
                            sw = DenseHead._sample_weights(self._targets(yo[n_held:]), self._step_class_weight(yo[n_held:]), None)
                    else:
                        sw = DenseHead._sample_weights(self._targets(yo[n_held:]), self._step_class_weight(yo[n_held:]), None)
                    if weights is None:
                        weights = sw is not None
                    elif weights != (sw is not None):
                        raise ValueError("class weights for some steps of a block and not for others")
                    need_f = n * od + (n - n_held if weights else 0)
                    if io + 2 * n > cap_i or fo + need_f > cap_f or n - n_held > 4096 or n_held > 4096:
                        raise ValueError("a generator batch of %d rows is too large for the indexed customTrainModel "
                                         "(set _index_steps = False)" % n)
                    inp[io:io + n] = li[order]
                    inp[io + n:io + 2 * n] = ri[order]
                    if base_targets:                            # to_categorical(y, 2): [1 - y, y]
                        t2 = fnp[fo:fo + 2 * n].reshape(n, 2)
                        t2[:, 1] = yo[:, 0]
                        t2[:, 0] = 1 - yo[:, 0]
                    else:
                        fnp[fo:fo + n * od] = np.asarray(self._targets(yo), dtype=np.float32).reshape(-1)
                    if weights:
                        fnp[fo + n * od:fo + need_f] = sw
                    desc[k] = (io, fo, n, n_held)
                    io += 2 * n
                    fo += need_f
                    k += 1
                run(st, desc, k, io, fo, bool(weights))
                done += k
                if ended is not None:
                    for s_ in sets:
                        collect(s_, sums)
                    raise ended
            # end of the epoch: both sets in step order (the older one first)
            collect(sets[turn % 2], sums)
            collect(sets[(turn + 1) % 2], sums)
            logs.append(tuple(sums / steps_per_epoch))
        return logs

    def maybeLoadFromMemory(self):
        try:
            self.siamese_net.load_weights(self.modelName + ".h5")
            return True
        except Exception:
            return False

    def save(self, customName=None):
        if not customName:
            self.siamese_net.save_weights(self.modelName + ".h5")
        else:
            self.siamese_net.save_weights(customName + ".h5")

    def preprocess(self, X):
        return X

    def predict(self, X):
        return self.siamese_net.predict(self.preprocess(X), batch_size=1024)


class SmallRes(SiameseNetwork, object):
    """code/siamese.py:134-184.  `preprocess` = (x - 128)/128 per side, applied by predict and finetune
    (and by customTrainModel only when preprocess=True, code/siamese.py:88-89)."""
    _identity_preprocess = False

    def __init__(self, imageShape, featureShape, name, learningRate, seed=None, adadelta_epsilon=1e-8):
        from .smallres import SmallResNet
        self.learningRate = learningRate
        self.shape = imageShape
        self.modelName = name
        self.siamese_net = SmallResNet(imageShape, featureShape[0], lr=learningRate, rho=0.95, eps=adadelta_epsilon,
                                       seed=seed)

    def getDenseBarebones(self):
        return [(128, 'relu'), (32, 'relu'), (2, None)]

    def preprocess(self, X):
        X_temp = [((x.float() if hasattr(x, "detach") else np.asarray(x, dtype=np.float32)) - 128.) / 128. for x in X]
        return X_temp

    def predict(self, X):
        return self.siamese_net.predict(self.preprocess(X), batch_size=1024)


class FaceVGG16:
    """code/siamese.py:187-200: VGGFace VGG-16 pool5 features (25088-d at 224 x 224).  `weights`: path of
    keras-vggface's `rcmalli_vggface_tf_notop_vgg16.h5`, a dict, or None for synthetic weights."""

    def __init__(self, shape, weights=None, dtype="bf16", max_batch=64, seed=1):
        from .vgg16 import VGGFace16
        self.shape = shape + (3,)
        self.model = VGGFace16(image_size=tuple(shape), weights=weights, dtype=dtype, max_batch=max_batch, seed=seed)

    def preprocess(self, X):
        """utils.preprocess_input(np.copy(X), version=1): RGB -> BGR, subtract the VGGFace channel means."""
        from .vgg16 import MEAN_BGR
        X_temp = np.ascontiguousarray(np.array(X, dtype=np.float32, copy=True)[..., ::-1])
        for c in range(3):
            X_temp[..., c] -= MEAN_BGR[c]
        return X_temp

    def process(self, X):
        if isinstance(X, (list, tuple)):
            X = np.stack(X)
        return self.model.predict(X, batch_size=128, preprocessed=False)


class RESNET50:
    """code/siamese.py:203-216: VGGFace2 ResNet-50 features (2048-d) at 224 x 224.  `weights`: path of
    the keras-vggface weight file (`rcmalli_vggface_tf_notop_resnet50.h5`), a dict, or None for
    synthetic weights (keras-vggface downloads its file; there is no network here)."""

    def __init__(self, shape, weights=None, dtype=None, max_batch=128, seed=1, screen_dtype=None):
        from .resnet50 import VGGResNet50
        self.shape = shape + (3,)
        # dtype=None: "f16x2", split precision — features to float32 accuracy, so that the selection the drivers make from
        # them (code/ALINK.py:67, code/ALINK_MTP.py:84, code/existing_al.py:58) follows the reference's float32 arithmetic;
        # "bf16" (3x faster, 1 - cos ~1e-4) and "f16" remain explicit choices
        self.model = VGGResNet50(image_size=tuple(shape), weights=weights, dtype=dtype or "f16x2", max_batch=max_batch, seed=seed)
        # screen_dtype ("bf16" | "f16"): a second handle on the same weights in a 16-bit mode, for screen-then-settle selection
        # (settle.py; the loop of alink_loop.py uses `process_screen` for the noisy copies when it exists).  Not in the reference.
        self.screen = None
        if screen_dtype:
            self.screen = VGGResNet50(image_size=tuple(shape), weights=weights, dtype=screen_dtype, max_batch=max_batch, seed=seed)
            self.process_screen = self._process_screen

    def _process_screen(self, X):
        if isinstance(X, (list, tuple)):
            X = np.stack(X)
        return self.screen.predict(X, batch_size=128, preprocessed=False)

    def preprocess(self, X):
        """utils.preprocess_input(np.copy(X), version=2): RGB -> BGR, subtract the VGGFace2 channel means."""
        from .resnet50 import MEAN_BGR
        X_temp = np.array(X, dtype=np.float32, copy=True)[..., ::-1]
        X_temp = np.ascontiguousarray(X_temp)
        X_temp[..., 0] -= MEAN_BGR[0]
        X_temp[..., 1] -= MEAN_BGR[1]
        X_temp[..., 2] -= MEAN_BGR[2]
        return X_temp

    def process(self, X):
        """model.predict(preprocess(X), batch_size=128); here the flip and mean subtraction run in the
        stem kernel's loader, so raw pixels go to the device once."""
        if isinstance(X, (list, tuple)):
            X = np.stack(X)
        return self.model.predict(X, batch_size=128, preprocessed=False)


class ArcFace:
    def __init__(self, shape, model_path, dtype=None, max_batch=292, enable_grad=False, small_batch_split=False,
                 gpu=None, screen_dtype="default"):
        # dtype=None: face_model.default_dtype — "f16x2" (selection sets identical to the reference's float32
        # arithmetic), or "bf16" when the gradient pass is requested
        args = _Args({
            "enable_grad": enable_grad,
            "small_batch_split": small_batch_split,   # latency mode for batches <= 32 (include/alink_hip.h)
            "image_size": "%d,%d" % (shape[0], shape[1]),
            "model": model_path + ",0",
            # the reference hard-codes gpu 0 (code/siamese.py:223) in a single-GPU process; here None = this
            # process's current device, so that rank k of a one-process-per-GPU job builds its model on GPU k
            "gpu": gpu,
            "threshold": 1.24,
            "dtype": dtype,
            "max_batch": max_batch,
        })
        self.model = face_model.FaceModel(args)
        # screen_dtype: the fast form for screen-then-settle selection (settle.py) — the bulk of the images goes through
        # `process_screen`, only those near a cut through `process`.  "f16" / "bf16": a second handle on the same
        # checkpoint in that 16-bit mode; "f16x2/1": the one-product form of the exact handle itself (no second copy, no
        # float16 range to leave, 8x finer than bf16 at ~0.8x its rate); "auto": f16 where the network's activations
        # fit its range (the fastest and finest), else f16x2/1 (exact model in split precision) or bf16.  Not in the reference.
        # Default: "auto" when the model is the exact one (split precision) — `process` and every result stay the exact mode's, the
        # framework loop just reaches them ~2x sooner; None / False builds no screening form.
        # The screening handle is built LAZILY, on the first process_screen / backbones() / .screen access: a caller that
        # only ever calls process() (the reference's API) pays for one network, not two.
        if screen_dtype == "default":
            screen_dtype = "auto" if (self.model.model.dtype == "f16x2" and not enable_grad) else None
        self._screen_dtype, self._screen_args, self._screen = (screen_dtype or None), args, None
        if self._screen_dtype:
            self.process_screen = self._process_screen

    @property
    def screen(self):
        """the screening form (None when the model has none); built on first use"""
        if self._screen is None and self._screen_dtype:
            screen_dtype, args, exact = self._screen_dtype, self._screen_args, self.model.model
            if screen_dtype == "auto":
                from ._abi import AlinkError
                cand = face_model.FaceModel(_Args(dict(args, dtype="f16")))
                try:                                   # do the probe images (uniform noise, black, white) fit plain f16?
                    cand.model.embed_device(cand.model._probe_images())
                    self._screen = cand
                except AlinkError:
                    del cand
                    screen_dtype = "f16x2/1" if exact.dtype == "f16x2" else "bf16"
            if self._screen is None:
                if screen_dtype == "f16x2/1":
                    view = exact.screening_view()
                    self._screen = _Args({"model": view, "get_features": view.embed})
                else:
                    self._screen = face_model.FaceModel(_Args(dict(args, dtype=screen_dtype)))
        return self._screen

    def search_handle(self, dtype):
        """A handle on the same checkpoint in `dtype` ("bf16" | "f16"), built on first use and kept: what the few-pixel
        attack's search="bf16" ranks its candidates with (attack._device_parts) — the search is a random one, which
        arithmetic orders its candidates is not contractual, and bf16 is the fastest form there is (48 k against the
        one-product screening form's 36 k and the exact mode's 16 k forwards/s at IR-100)."""
        if self.model.model.dtype == dtype:
            return self.model
        cache = self.__dict__.setdefault("_search_handles", {})
        if dtype not in cache:
            cache[dtype] = face_model.FaceModel(_Args(dict(self._screen_args, dtype=dtype)))
        return cache[dtype]

    def backbones(self):
        """(screening IRBackbone or None, exact IRBackbone): what distributed.committee_pool_topk_settled takes"""
        return (self.screen.model if self.screen is not None else None), self.model.model

    # The split-precision mode's calibration state travels with the checkpoint (the persistence contract of the
    # reference's models is save / maybeLoadFromMemory, code/siamese.py:114-125: a file beside the model, False when it
    # cannot be read): embeddings are bit-reproducible only under the same scales.
    def _calibration_path(self, path=None):
        if path:
            return path
        prefix = self.model.args.model.split(",")[0]
        if prefix.startswith("synthetic:"):
            raise ValueError("a synthetic checkpoint has no file to keep the calibration beside: pass a path")
        return prefix + ".alink_scales.json"

    def calibrate(self, X):
        """choose the split-precision scales from images like the ones to come (any dtype other than f16x2: no-op)"""
        if self.model.model.dtype == "f16x2":
            self.model.model.calibrate(np.stack(X) if isinstance(X, (list, tuple)) else X)

    def save_calibration(self, path=None):
        import json
        with open(self._calibration_path(path), "w") as f:
            json.dump(self.model.model.state(), f)

    def maybeLoadCalibration(self, path=None):
        import json
        try:
            with open(self._calibration_path(path)) as f:
                self.model.model.load_state(json.load(f))
            return True
        except Exception:
            return False

    def preprocess(self, X):
        return X

    def _process_screen(self, X):
        X = self.preprocess(X)
        if isinstance(X, (list, tuple)):
            X = np.stack(X)
        if len(X) == 0:
            return np.zeros((0, 512), dtype=np.float32)
        from ._abi import AlinkError
        try:
            return self.screen.get_features(X)
        except AlinkError:
            # a batch the 16-bit screening form cannot hold (plain f16 left its range): the exact mode's own embeddings are a
            # perfectly good "screening" of themselves — nothing near a cut will move when it is settled
            return self.model.get_features(X)

    def process(self, X):
        """(N,H,W,3) float RGB 0..255 -> (N,512).  The reference loops get_input/get_feature per
        image at batch 1 (code/siamese.py:234); here the whole array is one batched launch chain."""
        X = self.preprocess(X)
        if isinstance(X, (list, tuple)):
            X = np.stack(X)
        if len(X) == 0:
            return np.zeros((0, 512), dtype=np.float32)
        return self.model.get_features(X)

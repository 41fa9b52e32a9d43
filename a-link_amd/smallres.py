"""smallres — SmallRes, the end-to-end-trained low-resolution siamese CNN (reference
code/siamese.py:134-184; driver code/ALINK_MTP.py:107,121,255).

`SmallResNet` is the Keras-Model-like object (`siamese_net`): predict / train_on_batch / test_on_batch /
fit / get_weights / set_weights / save_weights / load_weights, all on libalink_hip.so (smallres.hip).
Dropout masks are drawn on the host with np.random (the reference's come from TensorFlow's op-level
RNG, which cannot be reproduced; the distribution — keep probability 0.75, scale 1/0.75 — is the same).
"""
import contextlib
import ctypes as C

import numpy as np

from . import _abi
from .head import KerasFitMixin, glorot_uniform

MAXN = 256
_NO_CONTEXT = contextlib.nullcontext()


class SmallResNet(KerasFitMixin):
    def __init__(self, image_shape, feat, lr=1.0, rho=0.95, eps=1e-8, seed=None, device=None, prescale=False):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _abi.AlinkError("no ROCm device visible: a-link_amd computes only on the GPU (no CPU fallback)")
        device = _abi.resolve_device(device)          # None: the current torch device
        self.device = "cuda:%d" % device
        self.lib = _abi.init(device)
        self.H, self.W = int(image_shape[0]), int(image_shape[1])
        assert int(image_shape[2]) == 3
        self.feat = int(feat)
        with _abi.on_device(device):
            self.h = self.lib.alink_smallres_create(self.H, self.W, self.feat, lr, rho, eps)
        if not self.h:
            raise _abi.AlinkError("alink_smallres_create: " + self.lib.alink_last_error().decode())
        self.lr = lr
        a, b = C.c_int(), C.c_int()
        _abi.check(self.lib.alink_smallres_mask_sizes(self.h, C.byref(a), C.byref(b)))
        self.mask_sizes = (a.value, b.value)
        self.flat = b.value
        self.training_dropout = True
        rng = np.random.RandomState(seed) if seed is not None else np.random
        ws = []
        for ci, co in ((3, 32), (32, 32), (32, 64), (64, 64)):
            lim = np.sqrt(6.0 / (9 * ci + 9 * co))               # glorot_uniform on (3,3,ci,co)
            ws += [rng.uniform(-lim, lim, (3, 3, ci, co)).astype(np.float32), np.zeros(co, np.float32)]
        ws += [glorot_uniform(rng, self.flat, self.feat), np.zeros(self.feat, np.float32)]
        ws += [glorot_uniform(rng, self.feat, 128), np.zeros(128, np.float32), glorot_uniform(rng, 128, 32),
               np.zeros(32, np.float32), glorot_uniform(rng, 32, 2), np.zeros(2, np.float32)]
        self.set_weights(ws)
        self._metrics = torch.zeros(2, dtype=torch.float32, device=self.device)
        # {loss, accuracy} of a step land in pinned host memory, written by the kernel itself (as DenseHead's do): a step ends with
        # one stream wait instead of a device-to-host copy
        self._metrics_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        self._metrics_np = np.zeros(2, dtype=np.float32)
        self._tdev = torch.device(self.device)
        self._tdev_index = self._tdev.index
        self._mask_buf = {}
        # Optional, OFF: train_on_batch can replay its ~40 launches as ONE captured graph (use_graph = True + alink_smallres_set_graph):
        # a stream of this model's own (torch's default stream is the NULL stream, which cannot be captured) and staging buffers that
        # keep the operands at stable addresses.  Measured on MI355X / ROCm 7.2: 0.457 ms per 16-pair step replayed against 0.375
        # as plain launches (tools/experiments/smallres_graph_ab.py) — a replayed node costs more than a launch here.
        self._stream = torch.cuda.Stream(device=self._tdev)
        self._stage = {}
        self._pin = {}
        self.use_graph = False
        self.prescale = 1 if prescale else 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.alink_smallres_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _shapes(self):
        return [(3, 3, 3, 32), (32,), (3, 3, 32, 32), (32,), (3, 3, 32, 64), (64,), (3, 3, 64, 64), (64,),
                (self.flat, self.feat), (self.feat,), (self.feat, 128), (128,), (128, 32), (32,), (32, 2), (2,)]

    def set_weights(self, ws):
        shapes = self._shapes()
        assert len(ws) == len(shapes)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(w, np.float32).reshape(s).ravel()
                                                    for w, s in zip(ws, shapes)]), dtype=np.float32)
        _abi.check(self.lib.alink_smallres_set_params(self.h, _abi.ptr(flat), flat.size), "alink_smallres_set_params")

    def get_weights(self):
        n = self.lib.alink_smallres_num_params(self.h)
        flat = np.empty(n, dtype=np.float32)
        _abi.check(self.lib.alink_smallres_get_params(self.h, _abi.ptr(flat), n), "alink_smallres_get_params")
        out, o = [], 0
        for s in self._shapes():
            k = int(np.prod(s))
            out.append(flat[o:o + k].reshape(s).copy())
            o += k
        return out

    def get_lr(self):
        return self.lr

    def set_lr(self, lr):
        self.lr = float(lr)
        _abi.check(self.lib.alink_smallres_set_lr(self.h, float(lr)))

    def save_weights(self, path):
        """Keras save_weights of the graph at reference code/siamese.py:139-168: the shared tower is
        one nested layer `sequential_1` holding conv2d_1..4 and dense_1; the head is dense_2..4."""
        ws = self.get_weights()
        if path.endswith(".npz"):
            np.savez(path, *ws)
            return
        from . import hdf5_lite
        tower = []
        for i, n in enumerate(["conv2d_1", "conv2d_2", "conv2d_3", "conv2d_4", "dense_1"]):
            tower += [("%s/kernel:0" % n, ws[2 * i]), ("%s/bias:0" % n, ws[2 * i + 1])]
        layers = [("input_1", []), ("input_2", []), ("sequential_1", tower), ("lambda_1", [])]
        for j, n in enumerate(["dense_2", "dense_3", "dense_4"]):
            layers.append((n, [("%s/kernel:0" % n, ws[10 + 2 * j]), ("%s/bias:0" % n, ws[11 + 2 * j])]))
        layers.append(("activation_7", []))
        hdf5_lite.save_keras_weights(path, layers)

    def load_weights(self, path):
        with open(path, "rb") as f:
            magic = f.read(4)
        if magic == b"PK\x03\x04":
            with np.load(path) as z:
                self.set_weights([z["arr_%d" % i] for i in range(len(self._shapes()))])
            return
        from . import hdf5_lite
        flat = [a for _, ws in hdf5_lite.load_keras_weights(path) for _, a in ws]
        shapes = self._shapes()
        if len(flat) != len(shapes) or any(tuple(a.shape) != tuple(s) for a, s in zip(flat, shapes)):
            raise ValueError("weight file does not match this SmallRes (%d tensors, expected %d)" % (len(flat), len(shapes)))
        self.set_weights(flat)

    def _dev(self, a):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a.to(self.device, torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def predict(self, X, batch_size=1024, verbose=0):
        as_torch = isinstance(X[0], self.torch.Tensor)
        L, R = self._dev(X[0]), self._dev(X[1])
        n = L.shape[0]
        out = self.torch.empty((n, 2), dtype=self.torch.float32, device=self.device)
        for s in range(0, n, MAXN):
            m = min(MAXN, n - s)
            _abi.check(self.lib.alink_smallres_forward(self.h, _abi.ptr(L[s:s + m]), _abi.ptr(R[s:s + m]), m,
                                                       self.prescale, _abi.ptr(out[s:s + m]), _abi.current_stream(self.device)),
                       "alink_smallres_forward")
        return out if as_torch else out.cpu().numpy()

    def draw_masks(self, n):
        """keep-masks (u8) for the 2n tower passes: Dropout(0.25) after each pool (code/siamese.py:146,153) — on the host, for callers
        that pass `masks=` explicitly (the parity tests hand the same masks to the oracle); train_on_batch draws its own on the device."""
        e1, e2 = self.mask_sizes
        return (np.random.rand(2 * n * e1 + 2 * n * e2) >= 0.25).astype(np.uint8)

    def _staged(self, key, a, dtype=None):
        """a device tensor holding `a` at an address that is the same from step to step (host array, or device tensor on any stream
        the caller has made this model's stream wait for)"""
        torch = self.torch
        dtype = dtype or torch.float32
        if isinstance(a, torch.Tensor):
            src = a
        else:
            src = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8 if dtype is torch.uint8 else np.float32))
        buf = self._stage.get((key, tuple(src.shape)))
        if buf is None:
            buf = self._stage[(key, tuple(src.shape))] = torch.empty(tuple(src.shape), dtype=dtype, device=self.device)
        buf.copy_(src, non_blocking=True)
        return buf

    def _up(self, key, a):
        """operand `a` as a float32 device tensor: device tensors as they are, a host array through a pinned staging buffer and an
        asynchronous copy (a pageable upload is a synchronous ~15 us each); the step ends with a stream synchronisation, so the
        buffer is free again by the next one."""
        torch = self.torch
        if isinstance(a, torch.Tensor):
            if a.dtype is torch.float32 and a.is_cuda and a.is_contiguous() and a.device.index == self._tdev_index:
                return a
            return a.to(self.device, torch.float32).contiguous()
        a = np.asarray(a, dtype=np.float32)
        buf = self._pin.get((key, a.shape))
        if buf is None:
            t = torch.empty(a.shape, dtype=torch.float32).pin_memory()
            buf = self._pin[(key, a.shape)] = (t, t.numpy(), torch.empty(a.shape, dtype=torch.float32, device=self.device))
        buf[1][...] = a
        buf[2].copy_(buf[0], non_blocking=True)
        return buf[2]

    def _up_all(self, arrays):
        """the host arrays of one step (images of both sides, targets, sample weights) through ONE pinned staging buffer and ONE
        asynchronous copy into a device buffer kept from step to step (three copies and three allocations until round 6):
        returns the device views, or None if any operand is not a host array (those go through _up one by one)."""
        torch = self.torch
        if any(isinstance(a, torch.Tensor) for a in arrays if a is not None):
            return None
        arrs = [None if a is None else np.asarray(a, dtype=np.float32) for a in arrays]
        key = tuple(None if a is None else a.shape for a in arrs)
        ent = self._pin.get(key)
        if ent is None:
            offs, o = [], 0
            for a in arrs:
                offs.append(o)
                o += 0 if a is None else (a.size + 63) // 64 * 64          # 256-byte aligned pieces
            host = torch.empty(max(o, 64), dtype=torch.float32).pin_memory()
            dev = torch.empty(max(o, 64), dtype=torch.float32, device=self.device)
            hv = host.numpy()
            views = [None if a is None else hv[f:f + a.size].reshape(a.shape) for a, f in zip(arrs, offs)]
            dviews = [None if a is None else dev[f:f + a.size].view(a.shape) for a, f in zip(arrs, offs)]
            ent = self._pin[key] = (host, dev, views, dviews, o)
        host, dev, views, dviews, o = ent
        for v, a in zip(views, arrs):
            if a is not None:
                v[...] = a
        dev.copy_(host, non_blocking=True)
        return dviews

    def train_on_batch(self, x, y, class_weight=None, sample_weight=None, masks=None):
        torch = self.torch
        n = len(y)
        assert n <= MAXN, "train batches larger than %d pairs are not supported" % MAXN
        sw = sample_weight
        if sw is None and class_weight is not None:
            sw = np.asarray([class_weight[c] for c in np.asarray(y).argmax(axis=1)], np.float32)
        cur = torch.cuda.current_stream(self._tdev)
        if masks is None and not self.use_graph and not isinstance(x[0], torch.Tensor) and not isinstance(x[1], torch.Tensor) \
                and not isinstance(y, torch.Tensor) and not isinstance(sw, torch.Tensor):
            # host operands, as the reference's Keras call hands them: ONE synchronous library call stages them (pinned memory of the
            # handle, one upload), draws the masks, runs the step, synchronises and returns {loss, accuracy}
            L, R = np.ascontiguousarray(x[0], np.float32), np.ascontiguousarray(x[1], np.float32)
            yh = np.ascontiguousarray(y, np.float32)
            swh = None if sw is None else np.ascontiguousarray(sw, np.float32)
            drop = 1 if self.training_dropout else 0
            seed = int(np.random.randint(0, 2 ** 31 - 1)) if drop else 0      # (one np.random draw per step, as in the device-operand form)
            _abi.check(self.lib.alink_smallres_train_on_batch_host(self.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yh), _abi.ptr(swh), n, self.prescale,
                                                                   drop, seed, _abi.ptr(self._metrics_np), C.c_void_p(cur.cuda_stream)),
                       "alink_smallres_train_on_batch_host")
            return self._metrics_np.tolist()
        st = self._stream if self.use_graph else cur
        if st is not cur:
            st.wait_stream(cur)                       # inputs produced on the caller's stream
        # (entering a stream context costs ~5 us of host time before the step's first launch, with the device idle: only when
        # the step runs on another stream than the caller's)
        with (torch.cuda.stream(st) if st is not cur else _NO_CONTEXT):
            if self.use_graph:
                L, R, yd = self._staged("L", x[0]), self._staged("R", x[1]), self._staged("y", y)
                swd = self._staged("sw", sw) if sw is not None else None
            else:
                staged = self._up_all((x[0], x[1], y, sw))
                if staged is not None:
                    L, R, yd, swd = staged
                else:
                    L, R, yd = self._up("L", x[0]), self._up("R", x[1]), self._up("y", y)
                    swd = self._up("sw", sw) if sw is not None else None
            if masks is None and self.training_dropout:
                # the keep-masks are drawn ON THE DEVICE (Philox, keyed by one 31-bit seed taken from np.random per step): drawing
                # 2n(e1 + e2) = 304,128 uniforms with np.random on the host was 0.74 of the step's 1.68 ms.  One np.random draw per step
                # keeps the ranks of a multi-rank loop in step (alink_loop.sync_host_randomness) like the host-drawn masks did.
                # They are drawn by the step's own first launch (alink_smallres_train_step_drawn: the bytes alink_keep_masks writes).
                e1, e2 = self.mask_sizes
                md = self._mask_buf.get(n)               # (a step's masks are consumed by that step, in stream order)
                if md is None:
                    md = self._mask_buf[n] = torch.empty(2 * n * (e1 + e2), dtype=torch.uint8, device=self.device)
                _abi.check(self.lib.alink_smallres_train_step_drawn(self.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yd), _abi.ptr(swd), n,
                                                                    self.prescale, _abi.ptr(md), int(np.random.randint(0, 2 ** 31 - 1)), 0.0, 1,
                                                                    C.c_void_p(self._metrics_host.data_ptr()), C.c_void_p(st.cuda_stream)),
                           "alink_smallres_train_step_drawn")
            else:
                if masks is not None:
                    md = self._staged("masks", masks, torch.uint8) if self.use_graph else \
                        torch.from_numpy(np.ascontiguousarray(masks, np.uint8)).to(self.device)
                else:
                    md = None
                _abi.check(self.lib.alink_smallres_train_step(self.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yd), _abi.ptr(swd), n,
                                                              self.prescale, _abi.ptr(md), 0.0, 1, C.c_void_p(self._metrics_host.data_ptr()),
                                                              C.c_void_p(st.cuda_stream)), "alink_smallres_train_step")
        st.synchronize()
        return self._metrics_host.tolist()

    # -- data-parallel step (distributed.dp_train_on_batch) --------------------------------------------
    # Rows from which the step is SHARDED in mode "auto".  One 16-pair step is 0.35 ms on one MI355X (profiles/r06d_*); its flat
    # gradient buffer is 20.2 MB at 32 x 32 / 2048 — an 8-rank exchange moves 7 x 20 MB into every rank over 7 xGMI links
    # (~150 GB/s each: >= 0.15 ms before any latency) plus the sum.  Replicas fed the same batch and the same mask seed stay
    # bit-identical, so below ~8 steps' worth of rows per rank replication wins; the threshold is that estimate — no multi-GPU
    # node was available to measure it — and `mode="sharded"` forces the exchange at any size.
    DP_SHARD_MIN_ROWS = 128

    def grads_tensor(self, with_metrics=False):
        """torch alias of the flat [tower | head] gradient buffer (for torch.distributed collectives); with_metrics appends
        the 4 spare floats that follow it (slot 0 / 1: loss, accuracy of a data-parallel step)."""
        torch = self.torch
        n = self.lib.alink_smallres_num_params(self.h) + (4 if with_metrics else 0)

        class _CAI(object):
            pass
        holder = _CAI()
        holder.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(self.lib.alink_smallres_grads_dev(self.h)), False),
                                           "version": 2, "strides": None}
        t = torch.as_tensor(holder, device=self.device)
        t._alink_owner = self
        return t

    def dp_begin(self, n, group):
        """the step's dropout seed: every rank draws one (np.random stays in step on all ranks, as in the replicated form)
        and rank 0's is the one used"""
        seed = int(np.random.randint(0, 2 ** 31 - 1)) if self.training_dropout else -1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            box = [seed]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            seed = int(box[0])
        return seed

    def _masks_of_rows(self, lo, hi, n, seed):
        """the keep-masks the whole-batch step (train_on_batch: one stream of 2n(e1 + e2) elements, [L ; R] image order per
        pool) draws for pair rows lo : hi, in the layout of a batch of hi - lo pairs"""
        e1, e2 = self.mask_sizes
        k = hi - lo
        md = self.torch.empty(2 * k * (e1 + e2), dtype=self.torch.uint8, device=self.device)
        st = _abi.current_stream(self.device)
        parts = ((0, lo * e1, k * e1), (k * e1, (n + lo) * e1, k * e1),                       # pool 1: L rows, R rows
                 (2 * k * e1, 2 * n * e1 + lo * e2, k * e2), (2 * k * e1 + k * e2, 2 * n * e1 + (n + lo) * e2, k * e2))
        for dst, first, count in parts:
            _abi.check(self.lib.alink_keep_masks_at(_abi.ptr(md[dst:dst + count]), count, 0.75, seed, first, st), "alink_keep_masks_at")
        return md

    def dp_local_grads(self, x, y, w_all, lo, hi, n, grad_scale, m, ctx):
        take = lambda a: a[lo:hi] if hasattr(a, "shape") else np.asarray(a)[lo:hi]
        L, R, yd = self._dev(take(x[0])), self._dev(take(x[1])), self._dev(take(y))
        swd = None if w_all is None else self._dev(w_all[lo:hi])
        k = hi - lo
        assert k <= MAXN, "a rank's slice of more than %d pairs is not supported" % MAXN
        md = self._masks_of_rows(lo, hi, n, ctx) if (self.training_dropout and ctx is not None and ctx >= 0) else None
        _abi.check(self.lib.alink_smallres_train_step(self.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yd), _abi.ptr(swd), k,
                                                      self.prescale, _abi.ptr(md), grad_scale, 0, _abi.ptr(m),
                                                      _abi.current_stream(self.device)), "alink_smallres_train_step")

    def dp_apply(self):
        _abi.check(self.lib.alink_smallres_apply_update(self.h, _abi.current_stream(self.device)), "alink_smallres_apply_update")

    def test_on_batch(self, x, y):
        n = len(y)
        if n <= MAXN:
            # one evaluation launch chain: operands through the step's staging (one upload), {loss, accuracy} written by the
            # device into pinned host memory, one stream synchronisation (a .to() per operand and a .cpu() of the metrics were a
            # third of the call)
            torch = self.torch
            staged = self._up_all((x[0], x[1], y, None))
            if staged is not None:
                L, R, yd, _ = staged
            else:
                L, R, yd = self._up("L", x[0]), self._up("R", x[1]), self._up("y", y)
            st = torch.cuda.current_stream(self._tdev)
            _abi.check(self.lib.alink_smallres_eval(self.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yd), n, self.prescale,
                                                    C.c_void_p(self._metrics_host.data_ptr()), C.c_void_p(st.cuda_stream)), "alink_smallres_eval")
            st.synchronize()
            return self._metrics_host.tolist()[:2]
        L, R, yd = self._dev(x[0]), self._dev(x[1]), self._dev(y)
        tot, seen = np.zeros(2), 0
        for s in range(0, L.shape[0], MAXN):
            m = min(MAXN, L.shape[0] - s)
            _abi.check(self.lib.alink_smallres_eval(self.h, _abi.ptr(L[s:s + m]), _abi.ptr(R[s:s + m]),
                                                    _abi.ptr(yd[s:s + m]), m, self.prescale, _abi.ptr(self._metrics),
                                                    _abi.current_stream(self.device)), "alink_smallres_eval")
            tot += self._metrics.cpu().numpy() * m
            seen += m
        return [float(tot[0] / seen), float(tot[1] / seen)]

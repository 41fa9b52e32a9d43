"""IRBackbone — host-side owner of one alink_backbone_t handle (include/alink_hip.h).

Device memory (inputs, outputs, workspace) is allocated through torch; the arithmetic runs in
libalink_hip.so.  Mirrors what face_model.get_model builds (reference code/face_model.py:28-41) but
binds a *batched* executor: the reference binds batch = 1 (code/face_model.py:39).
"""
import ctypes as C

import numpy as np

from . import _abi
from . import weights as W


class IRBackbone(object):
    DTYPES = {"bf16": _abi.DT_BF16, "f16": _abi.DT_F16, "f32": _abi.DT_F32, "f16x2": _abi.DT_F16X2}

    def __init__(self, params, image_size=(112, 112), emb=512, dtype="bf16", device=None, max_batch=292,
                 widths=W.WIDTHS, streams=2, shards_per_call=None, bn_eps=2e-5, enable_grad=False,
                 small_batch_split=False, lazy_range_check=False):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _abi.AlinkError("no ROCm device visible: a-link_amd computes only on the GPU (no CPU fallback)")
        device = _abi.resolve_device(device)          # None: the current torch device
        self.device = device
        self.lib = _abi.init(device)
        self.units = W.infer_units(params)
        self.emb = emb
        self.image_size = tuple(image_size)
        # 292 images per launch: 292 x 196 pixels / 224 per workgroup x 2 channel halves = 511 workgroups of the
        # 14-wide convolutions for the chip's 512 slots (1022 / 1024 at 28 wide) — a lone 256-image launch leaves 64 idle
        self.max_batch = int(max_batch)
        cfg = _abi.IRCfg()
        cfg.units[:] = list(self.units)
        cfg.widths[:] = list(widths)
        cfg.height, cfg.width = int(image_size[0]), int(image_size[1])
        cfg.emb = emb
        cfg.bn_eps = float(bn_eps)
        self.grad_enabled = bool(enable_grad)
        # dtype "auto": float16 storage (11 significant bits: 1 - cos ~4e-6 against the f32 arithmetic at IR-100 depth,
        # and selection sets that follow it: DESIGN.md §5) when the network's activations fit its range, else bfloat16
        # (8 significant bits, f32's range: 1 - cos ~3e-4).  The range is probed on three images at build time (uniform
        # noise, all 0, all 255); later inputs that leave it raise AlinkError in embed (never NaN embeddings).
        #
        # dtype "f16x2" — SPLIT PRECISION, the mode for selection: every activation and folded weight is an f16 pair
        # hi + lo (22 significant bits) under a power-of-two scale per tensor, three products on the f16 matrix cores
        # into f32 accumulators.  The accuracy of the float32 mode (active-learning selection sets identical to the f32
        # arithmetic: DESIGN.md §5) at about a third of the bf16 rate.  Scales are calibrated on the same three probe
        # images plus whatever `calibrate()` is given later.  For FIXED scales an image embeds to the same bits whatever
        # batch it arrives in; a different calibration moves embeddings by ~2e-7 (the lo halves of values 2^13 below a
        # tensor's maximum are subnormal f16 and round differently) — far below the mode's error, not zero: share scales
        # with state() / load_state().  A batch that leaves the range (32x above the calibration images' largest
        # activation) is re-calibrated on (every chunk of it) and re-run, which changes the scales from then on.
        self._shards_fixed = None if shards_per_call is None else int(shards_per_call)
        self.lazy_range_check = bool(lazy_range_check)
        order = ["f16", "bf16"] if dtype == "auto" else [dtype]
        for dt in order:
            cfg.dtype = self.DTYPES[dt]
            self.dtype = dt
            self._build(cfg, params, small_batch_split, enable_grad)
            if dt == "f16x2":
                self._ws = {}
                self.calibrate(self._probe_images())
            if dtype != "auto" or dt == "bf16" or self._range_probe_ok():
                break
            self.lib.alink_backbone_destroy(self.h)
            self.h = None
            self._ws = {}
        # shards_per_call: image shards ONE alink_embed call is split into on the library's internal streams.  None
        # (default) = 2 for a call that is alone on its stream and has >= 192 images — two half-launches de-phase each
        # other's prologue / epilogue bursts: IR-50, one 256-image batch per step 65.9k -> 72.0k embeddings/s, IR-100 at
        # 292 images +1.5 % — and 1 for the chunks of a multi-stream call, which already overlap (sharding those costs 11 %).
        self._set_shards(self._shards_fixed or 1)
        # Inputs larger than max_batch are cut into max_batch-image chunks issued round-robin on
        # `streams` side streams (each with its own workspace) and joined once at the end: chunks are
        # independent, and de-synchronising them lets the HBM bursts of one chunk's tiles overlap the
        # matrix-core phases of the others (+8...10 % measured on r100).  Two streams since the front of the network is
        # one persistent launch that owns every CU while it runs (front_c64.hip): 46.1 k against 44.8 k embeddings/s with
        # four (rounds 1-2, tile kernels throughout: four).
        self.n_streams = max(1, int(streams))
        self._side = None
        self._upload = None
        self._ws = {}

    def _build(self, cfg, params, small_batch_split, enable_grad):
        with _abi.on_device(self.device):             # the handle lives on the device current at create
            self.h = self.lib.alink_backbone_create(C.byref(cfg))
        if not self.h:
            raise _abi.AlinkError("alink_backbone_create: " + self.lib.alink_last_error().decode())
        if small_batch_split:      # latency mode: batches <= 32 split their convolutions over K (not bit-equal to fused)
            _abi.check(self.lib.alink_backbone_set_small_batch_split(self.h, 1), "alink_backbone_set_small_batch_split")
        if enable_grad:
            _abi.check(self.lib.alink_backbone_enable_grad(self.h), "alink_backbone_enable_grad")
        n = self.lib.alink_backbone_num_tensors(self.h)
        name, cnt = C.c_char_p(), C.c_size_t()
        for i in range(n):
            _abi.check(self.lib.alink_backbone_tensor_info(self.h, i, C.byref(name), C.byref(cnt)))
            key = name.value.decode()
            if key not in params:
                raise KeyError("checkpoint is missing tensor %s" % key)
            a = np.ascontiguousarray(params[key], dtype=np.float32)
            _abi.check(self.lib.alink_backbone_load(self.h, name.value, _abi.ptr(a), a.size), "load " + key)
        _abi.check(self.lib.alink_backbone_finalize(self.h), "alink_backbone_finalize")
        self._shards_now = None                       # a new handle: its shard count is not set yet

    def _probe_images(self):
        torch = self.torch
        h, w = self.image_size
        g = torch.Generator(device="cpu").manual_seed(0)
        return torch.stack([torch.randint(0, 256, (h, w, 3), generator=g).float(), torch.zeros(h, w, 3),
                            torch.full((h, w, 3), 255.0)]).to("cuda:%d" % self.device)

    def calibrate(self, x, merge=False):
        """dtype 'f16x2': choose the per-tensor power-of-two scales from these images (CUDA tensor or host array in any
        accepted layout; ALL of them, max_batch at a time: the first chunk sets the scales unless merge, every further
        chunk only lowers them).  merge=True only ever lowers a scale.  Embeddings are bit-reproducible for FIXED
        scales: share them with state() / load_state() (checkpoints, the ranks of a job)."""
        if self.dtype != "f16x2":
            raise _abi.AlinkError("only dtype='f16x2' is calibrated")
        torch = self.torch
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x if x.dtype == np.uint8 else x.astype(np.float32)))
        layout = self._layout_of(x, self.image_size)
        for i in range(0, x.shape[0], self.max_batch):
            xc = x[i:i + self.max_batch].to("cuda:%d" % self.device).contiguous()
            n = xc.shape[0]
            ws, wsb = self._workspace(n)
            torch.cuda.synchronize(self.device)
            _abi.check(self.lib.alink_backbone_calibrate(self.h, _abi.ptr(xc), layout, n, C.c_void_p(ws), wsb,
                                                         1 if (merge or i) else 0, _abi.current_stream(self.device)),
                       "alink_backbone_calibrate")

    def set_products(self, n):
        """dtype 'f16x2': matrix-core products per multiplication for the launches that follow — 3 = the exact mode
        (default), 1 = X_hi W_hi alone, the screening form on this same handle (include/alink_hip.h)."""
        _abi.check(self.lib.alink_backbone_set_products(self.h, int(n)), "alink_backbone_set_products")

    def screening_view(self):
        """An object with this backbone's embed / embed_device that runs them in the ONE-product screening form (same
        weights, scales and workspace; products are switched back to 3 after every call).  What screen-then-settle
        takes as its screening model when the exact model is this handle: no second copy of the network, no float16
        range to leave."""
        if self.dtype != "f16x2":
            raise _abi.AlinkError("screening_view is the one-product form of dtype='f16x2'")
        return _ScreeningView(self)

    def state(self):
        """The calibration state of the split-precision mode as a small dict of plain ints ({} for any other dtype): what
        has to travel with a checkpoint, and from the rank that calibrated to every other rank, for embeddings to be
        bit-identical there (include/alink_hip.h: alink_backbone_get_scales)."""
        n = self.lib.alink_backbone_num_scales(self.h)
        if n == 0:
            return {}
        e = (C.c_int * n)()
        _abi.check(self.lib.alink_backbone_get_scales(self.h, e, n), "alink_backbone_get_scales")
        return {"dtype": self.dtype, "units": [int(u) for u in self.units], "image_size": [int(v) for v in self.image_size],
                "scale_exponents": [int(v) for v in e]}

    def load_state(self, st):
        """Install scales saved by state() (same architecture and image size)."""
        if not st:
            return
        if st.get("dtype") != self.dtype or list(st.get("units", [])) != [int(u) for u in self.units] or \
                list(st.get("image_size", [])) != [int(v) for v in self.image_size]:
            raise _abi.AlinkError("calibration state of a %s %s network at %s does not fit this %s %s network at %s"
                                  % (st.get("dtype"), st.get("units"), st.get("image_size"), self.dtype, list(self.units), list(self.image_size)))
        v = [int(x) for x in st["scale_exponents"]]
        e = (C.c_int * len(v))(*v)
        _abi.check(self.lib.alink_backbone_set_scales(self.h, e, len(v)), "alink_backbone_set_scales")

    def _range_probe_ok(self):
        torch = self.torch
        x = self._probe_images()
        saved, self.dtype = self.dtype, "probe"       # _checked must not raise here
        self._ws, self._side, self._upload, self.n_streams = {}, None, None, 1
        out = self.embed_device(x)
        self.dtype = saved
        return bool(torch.isfinite(out).all())

    def _set_shards(self, n):
        if n != self._shards_now:
            _abi.check(self.lib.alink_backbone_set_streams(self.h, int(n)), "alink_backbone_set_streams")
            self._shards_now = n

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.alink_backbone_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- workspace -------------------------------------------------------------------------------
    def _workspace(self, n, slot=0):
        cur = self._ws.get(slot)
        if cur is None or n > cur[1]:
            nbytes = self.lib.alink_backbone_workspace_bytes(self.h, n)
            cur = (self.torch.empty(nbytes + 256, dtype=self.torch.uint8, device="cuda:%d" % self.device), n)
            self._ws[slot] = cur
        t = cur[0]
        off = (-t.data_ptr()) % 256
        return t.data_ptr() + off, t.numel() - off

    @staticmethod
    def _layout_of(x, image_size):
        if x.ndim != 4:
            raise ValueError("expected a 4-d batch of images, got shape %s" % (tuple(x.shape),))
        h, w = image_size
        is_u8 = str(x.dtype) in ("uint8", "torch.uint8")
        if tuple(x.shape[1:]) == (h, w, 3):
            return _abi.LAYOUT_NHWC_U8 if is_u8 else _abi.LAYOUT_NHWC_F32
        if tuple(x.shape[1:]) == (3, h, w) and not is_u8:
            return _abi.LAYOUT_NCHW_F32
        raise ValueError("images of shape %s do not match (N,%d,%d,3) / (N,3,%d,%d)" % (tuple(x.shape), h, w, h, w))

    def range_left(self, reset=True):
        """True if, since the last reset, some embedding came out non-finite (float16 storage has 5 exponent bits).  The
        flag is one word of pinned host memory that the last kernel of a forward writes: synchronise first."""
        return bool(self.lib.alink_backbone_range_flag(self.h, 1 if reset else 0))

    def check_range(self):
        """Synchronise this device and raise if a float16-storage forward left the range since the last check."""
        if self.dtype in ("f16", "f16x2"):
            self.torch.cuda.synchronize(self.device)
            if self.range_left():
                raise _abi.AlinkError("activations exceeded the float16 range in this network: build the backbone with "
                                      "dtype='bf16' (or, for dtype='f16x2', calibrate() on images like these)")

    def _checked(self, out, redo=None):
        """float16 storage: a network whose activations leave +-65504 (deep nets with synthetic weights do) comes out as
        NaN.  Fail loudly instead of returning it (bf16 has the range).  The test is a flag the FC-finish kernel raises,
        read after ONE synchronisation per call — or, with lazy_range_check, only in check_range() and at the start of
        the next call.  Split precision re-calibrates on the offending batch (scales only go down) and re-runs once."""
        if self.dtype not in ("f16", "f16x2") or self.lazy_range_check:
            return out
        self.torch.cuda.synchronize(self.device)
        if self.range_left():
            if self.dtype == "f16x2" and redo is not None:
                self.calibrate(redo, merge=True)
                out = self.embed_device(redo, out=out, _retry=False)
                return out
            raise _abi.AlinkError("activations exceeded the float16 range in this network: build the backbone with dtype='bf16'")
        return out

    def embed_device(self, x, out=None, _retry=True):
        """x: CUDA tensor (N,H,W,3) f32|u8 or (N,3,H,W) f32, contiguous.  Returns (N, emb) f32 CUDA."""
        torch = self.torch
        layout = self._layout_of(x, self.image_size)
        if self.lazy_range_check and self.dtype in ("f16", "f16x2") and self.range_left(reset=False):
            self.range_left()
            raise _abi.AlinkError("an earlier forward left the float16 range (lazy_range_check): its embeddings are not finite")
        if not x.is_contiguous():
            x = x.contiguous()
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, self.emb), dtype=torch.float32, device=x.device)
        nchunks = (n + self.max_batch - 1) // self.max_batch
        if nchunks == 1 or self.n_streams == 1:
            st = _abi.current_stream(self.device)
            for i in range(0, n, self.max_batch):
                m = min(self.max_batch, n - i)
                self._set_shards(self._shards_fixed or (2 if m >= 192 else 1))
                ws, wsb = self._workspace(m)
                _abi.check(self.lib.alink_embed(self.h, _abi.ptr(x[i:i + m]), layout, m, _abi.ptr(out[i:i + m]),
                                                C.c_void_p(ws), wsb, st), "alink_embed")
            return self._checked(out, x if _retry else None)
        if self._side is None:
            self._side = [torch.cuda.Stream(device=x.device) for _ in range(self.n_streams)]
        cur = torch.cuda.current_stream(x.device)
        ready = torch.cuda.Event()
        ready.record(cur)                       # inputs (and `out`) are valid in caller-stream order
        used = min(self.n_streams, nchunks)
        self._set_shards(self._shards_fixed or 1)
        for s in self._side[:used]:
            s.wait_event(ready)
        for j, i in enumerate(range(0, n, self.max_batch)):
            m = min(self.max_batch, n - i)
            slot = j % used
            ws, wsb = self._workspace(m, slot + 1)
            _abi.check(self.lib.alink_embed(self.h, _abi.ptr(x[i:i + m]), layout, m, _abi.ptr(out[i:i + m]),
                                            C.c_void_p(ws), wsb, C.c_void_p(self._side[slot].cuda_stream)),
                       "alink_embed")
        for s in self._side[:used]:
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)                # results are valid in caller-stream order
        return self._checked(out, x if _retry else None)

    # -- input gradient (FGSM / PGD extension) --------------------------------------------------------
    def _grad_workspace(self, n):
        cur = self._ws.get("grad")
        if cur is None or n > cur[1]:
            nbytes = self.lib.alink_backbone_grad_workspace_bytes(self.h, n)
            cur = (self.torch.empty(nbytes + 256, dtype=self.torch.uint8, device="cuda:%d" % self.device), n)
            self._ws["grad"] = cur
        t = cur[0]
        off = (-t.data_ptr()) % 256
        return t.data_ptr() + off, t.numel() - off

    def embed_with_cache(self, x):
        """Forward on <= max_batch images keeping what input_gradient needs.  x: CUDA float32 NHWC/NCHW."""
        if not self.grad_enabled:
            raise _abi.AlinkError("IRBackbone was built without enable_grad=True")
        torch = self.torch
        layout = self._layout_of(x, self.image_size)
        if layout == _abi.LAYOUT_NHWC_U8:
            raise ValueError("gradients need float32 pixels")
        x = x.contiguous()
        n = x.shape[0]
        assert 0 < n <= self.max_batch
        out = torch.empty((n, self.emb), dtype=torch.float32, device=x.device)
        ws, wsb = self._grad_workspace(n)
        _abi.check(self.lib.alink_embed_cached(self.h, _abi.ptr(x), layout, n, _abi.ptr(out), C.c_void_p(ws), wsb,
                                               _abi.current_stream(self.device)), "alink_embed_cached")
        self._cached = (n, layout, out)
        return out

    def input_gradient(self, demb):
        """d(loss)/d(pixels) for the batch of the last embed_with_cache, given demb = d(loss)/d(embedding)
        (n, emb) float32 CUDA.  Same layout as the forward input."""
        torch = self.torch
        n, layout, emb = self._cached
        demb = demb.to(torch.float32).contiguous()
        assert tuple(demb.shape) == (n, self.emb)
        h, w = self.image_size
        shape = (n, h, w, 3) if layout == _abi.LAYOUT_NHWC_F32 else (n, 3, h, w)
        dpix = torch.empty(shape, dtype=torch.float32, device=demb.device)
        ws, wsb = self._grad_workspace(n)
        _abi.check(self.lib.alink_embed_input_grad(self.h, _abi.ptr(demb), _abi.ptr(emb), layout, n, _abi.ptr(dpix),
                                                   C.c_void_p(ws), wsb, _abi.current_stream(self.device)), "alink_embed_input_grad")
        return dpix

    def embed(self, x):
        """numpy in -> numpy out (the reference's calling convention: host arrays, code/siamese.py:234);
        torch CUDA tensor in -> torch CUDA tensor out."""
        torch = self.torch
        if isinstance(x, np.ndarray):
            if x.dtype != np.uint8:
                x = np.ascontiguousarray(x, dtype=np.float32)
            x = np.ascontiguousarray(x)
            dev = "cuda:%d" % self.device
            n = x.shape[0]
            group = self.max_batch * self.n_streams
            if n <= group:
                return self.embed_device(torch.from_numpy(x).to(dev)).cpu().numpy()
            # Large host arrays go up in groups of max_batch * streams images: the (host-blocking) upload of a
            # group overlaps the launches already queued for the one before, and the device never holds
            # more than two groups of pixels (a 100k-image pool is 15 GB of float32).
            out = torch.empty((n, self.emb), dtype=torch.float32, device=dev)
            if self._upload is None:
                self._upload = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(self.device)
            for i in range(0, n, group):
                with torch.cuda.stream(self._upload):        # not ordered behind the compute already queued
                    xd = torch.from_numpy(x[i:i + group]).to(dev)
                    up = torch.cuda.Event()
                    up.record(self._upload)
                cur.wait_event(up)
                xd.record_stream(cur)                         # allocated on the upload stream, consumed in `cur` order
                self.embed_device(xd, out=out[i:i + group])
            return out.cpu().numpy()
        return self.embed_device(x)

    def profile(self, x):
        """One profiled forward: list of (kind, ms, flops) per launch (HIP events on the stream)."""
        torch = self.torch
        layout = self._layout_of(x, self.image_size)
        n = x.shape[0]
        assert n <= self.max_batch
        out = torch.empty((n, self.emb), dtype=torch.float32, device=x.device)
        ws, wsb = self._workspace(n)
        cap = 1024
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        kd = (C.c_int * cap)()
        nl = C.c_int(cap)
        _abi.check(self.lib.alink_embed_profile(self.h, _abi.ptr(x), layout, n, _abi.ptr(out), C.c_void_p(ws), wsb,
                                                _abi.current_stream(self.device), ms, fl, kd, C.byref(nl)), "alink_embed_profile")
        return [(kd[i], ms[i], fl[i]) for i in range(nl.value)]


class _ScreeningView(object):
    def __init__(self, bb):
        self.bb = bb
        self.dtype = "f16x2/1"
        self.max_batch, self.image_size, self.emb, self.device = bb.max_batch, bb.image_size, bb.emb, bb.device

    def _run(self, fn, *a, **k):
        self.bb.set_products(1)
        try:
            return fn(*a, **k)
        finally:
            self.bb.set_products(3)

    def embed_device(self, x, out=None):
        return self._run(self.bb.embed_device, x, out=out)

    def embed(self, x):
        return self._run(self.bb.embed, x)

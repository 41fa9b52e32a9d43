"""VGGResNet50 — host-side owner of one alink_resnet50_t handle: the keras-vggface ResNet-50 the
reference builds at code/siamese.py:203-209, with the slice of the Keras Model API it uses (`predict`).

Weights: a Keras weight file (the `rcmalli_vggface_tf_notop_resnet50.h5` keras-vggface downloads; read
with hdf5_lite.py), a dict of arrays with Keras names, or synthetic (He-normal convs, BN statistics
as in weights.synthetic_ir_params) when nothing is given — there is no network here to fetch the
pretrained file.
"""
import ctypes as C

import numpy as np

from . import _abi

UNITS = (3, 4, 6, 3)
MID = (64, 128, 256, 512)
MEAN_BGR = (91.4953, 103.8827, 131.0912)


def tensor_shapes():
    """Ordered {name: shape}: "<layer>/kernel" (kh, kw, in, out) and "<layer>/bn/<stat>" (out,)."""
    t = {}

    def conv(name, k, cin, cout):
        t[name + "/kernel"] = (k, k, cin, cout)
        for s in ("gamma", "beta", "moving_mean", "moving_variance"):
            t[name + "/bn/" + s] = (cout,)
    conv("conv1/7x7_s2", 7, 3, 64)
    cin = 64
    for s in range(4):
        mid, out = MID[s], 4 * MID[s]
        for u in range(1, UNITS[s] + 1):
            p = "conv%d_%d_" % (s + 2, u)
            conv(p + "1x1_reduce", 1, cin, mid)
            conv(p + "3x3", 3, mid, mid)
            conv(p + "1x1_increase", 1, mid, out)
            if u == 1:
                conv(p + "1x1_proj", 1, cin, out)
            cin = out
    return t


def synthetic_params(seed=1):
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in tensor_shapes().items():
        if name.endswith("/kernel"):
            v = rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))
        elif name.endswith("gamma"):
            # the last BN of a unit small, as trained residual nets have it: keeps activations bounded over 16 units
            v = rng.uniform(0.2, 0.5, shape) if "increase" in name else rng.uniform(0.5, 1.5, shape)
        elif name.endswith("beta") or name.endswith("moving_mean"):
            v = rng.standard_normal(shape) * 0.1
        else:
            v = rng.uniform(0.5, 1.5, shape)
        p[name] = np.ascontiguousarray(v, dtype=np.float32)
    return p


def load_keras_h5(path):
    """Keras weight file -> {"<layer>/kernel": ..., "<layer>/bn/gamma": ...} (the C library's names)."""
    from . import hdf5_lite
    out = {}
    for lname, ws in hdf5_lite.load_keras_weights(path):
        for wname, arr in ws:
            leaf = wname.rsplit("/", 1)[-1].split(":")[0]      # kernel | gamma | beta | moving_mean | moving_variance
            out[lname + "/" + leaf] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def save_keras_h5(path, params):
    from . import hdf5_lite
    layers = []
    for name in tensor_shapes():
        if name.endswith("/kernel"):
            l = name[:-len("/kernel")]
            layers.append((l, [(l + "/kernel:0", params[name])]))
            layers.append((l + "/bn", [(l + "/bn/%s:0" % s, params[l + "/bn/" + s])
                                       for s in ("gamma", "beta", "moving_mean", "moving_variance")]))
    hdf5_lite.save_keras_weights(path, layers)


class VGGResNet50(object):
    def __init__(self, image_size=(224, 224), weights=None, dtype="bf16", device=None, max_batch=128, bn_eps=1e-3,
                 seed=1):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _abi.AlinkError("no ROCm device visible: a-link_amd computes only on the GPU (no CPU fallback)")
        device = _abi.resolve_device(device)          # None: the current torch device
        self.device = device
        self.lib = _abi.init(device)
        self.image_size = tuple(image_size)
        self.max_batch = int(max_batch)
        if weights is None:
            params = synthetic_params(seed)
        elif isinstance(weights, str):
            params = load_keras_h5(weights)
        else:
            params = weights
        # dtype "f16x2": split precision (f16 pairs, three products on the f16 matrix cores, calibrated power-of-two scales:
        # a-link_amd/backbone.py) — features to float32 accuracy, what selection through this feature model needs
        self.dtype = dtype
        with _abi.on_device(device):
            self.h = self.lib.alink_resnet50_create(int(image_size[0]), int(image_size[1]),
                                                    {"bf16": _abi.DT_BF16, "f16": _abi.DT_F16, "f16x2": _abi.DT_F16X2}[dtype],
                                                    float(bn_eps))
        if not self.h:
            raise _abi.AlinkError("alink_resnet50_create: " + self.lib.alink_last_error().decode())
        name, cnt = C.c_char_p(), C.c_size_t()
        for i in range(self.lib.alink_resnet50_num_tensors(self.h)):
            _abi.check(self.lib.alink_resnet50_tensor_info(self.h, i, C.byref(name), C.byref(cnt)))
            key = name.value.decode()
            if key not in params:
                raise KeyError("weights are missing tensor %s" % key)
            a = np.ascontiguousarray(params[key], dtype=np.float32)
            _abi.check(self.lib.alink_resnet50_load(self.h, name.value, _abi.ptr(a), a.size), "load " + key)
        _abi.check(self.lib.alink_resnet50_finalize(self.h), "alink_resnet50_finalize")
        self._ws = None
        if dtype == "f16x2":         # scales from three probe images (uniform noise, black, white); a batch that leaves the
            h, w = self.image_size   # range later is re-calibrated on (scales only go down) and re-run
            g = torch.Generator(device="cpu").manual_seed(0)
            probe = torch.stack([torch.randint(0, 256, (h, w, 3), generator=g).float(), torch.zeros(h, w, 3),
                                 torch.full((h, w, 3), 255.0)])
            self.calibrate(probe)

    def calibrate(self, x, preprocessed=False, merge=False):
        """dtype 'f16x2': choose the per-tensor power-of-two scales from these images (all of them, max_batch at a time;
        every chunk after the first only lowers a scale, like merge=True)."""
        if self.dtype != "f16x2":
            raise _abi.AlinkError("only dtype='f16x2' is calibrated")
        torch = self.torch
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        for i in range(0, x.shape[0], self.max_batch):
            xc = x[i:i + self.max_batch].to("cuda:%d" % self.device).to(torch.float32).contiguous()
            ws, wsb = self._workspace(xc.shape[0])
            torch.cuda.synchronize(self.device)
            _abi.check(self.lib.alink_resnet50_calibrate(self.h, _abi.ptr(xc), xc.shape[0], 1 if preprocessed else 0, C.c_void_p(ws), wsb,
                                                         1 if (merge or i) else 0, _abi.current_stream(self.device)), "alink_resnet50_calibrate")

    def state(self):
        """calibration state of the split-precision mode ({} otherwise): see IRBackbone.state"""
        n = self.lib.alink_resnet50_num_scales(self.h)
        if n == 0:
            return {}
        e = (C.c_int * n)()
        _abi.check(self.lib.alink_resnet50_get_scales(self.h, e, n), "alink_resnet50_get_scales")
        return {"dtype": self.dtype, "image_size": [int(v) for v in self.image_size], "scale_exponents": [int(v) for v in e]}

    def load_state(self, st):
        if not st:
            return
        if st.get("dtype") != self.dtype or list(st.get("image_size", [])) != [int(v) for v in self.image_size]:
            raise _abi.AlinkError("calibration state does not fit this network")
        v = [int(x) for x in st["scale_exponents"]]
        e = (C.c_int * len(v))(*v)
        _abi.check(self.lib.alink_resnet50_set_scales(self.h, e, len(v)), "alink_resnet50_set_scales")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.alink_resnet50_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _workspace(self, n):
        if self._ws is None or n > self._ws[1]:
            nbytes = self.lib.alink_resnet50_workspace_bytes(self.h, n)
            self._ws = (self.torch.empty(nbytes + 256, dtype=self.torch.uint8, device="cuda:%d" % self.device), n)
        t = self._ws[0]
        off = (-t.data_ptr()) % 256
        return t.data_ptr() + off, t.numel() - off

    def embed_device(self, x, preprocessed=False, out=None, _retry=True):
        """x: CUDA (N, H, W, 3) float32 — raw RGB 0..255, or preprocess()'d when preprocessed=True."""
        torch = self.torch
        if x.ndim != 4 or tuple(x.shape[1:]) != self.image_size + (3,):
            raise ValueError("expected images of shape (N,%d,%d,3), got %s" % (self.image_size + (tuple(x.shape),)))
        x = x.to(torch.float32).contiguous()
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, 2048), dtype=torch.float32, device=x.device)
        for i in range(0, n, self.max_batch):
            m = min(self.max_batch, n - i)
            ws, wsb = self._workspace(m)
            _abi.check(self.lib.alink_resnet50_embed(self.h, _abi.ptr(x[i:i + m]), m, 1 if preprocessed else 0,
                                                     _abi.ptr(out[i:i + m]), C.c_void_p(ws), wsb, _abi.current_stream(self.device)),
                       "alink_resnet50_embed")
        if self.dtype in ("f16", "f16x2"):       # 16-bit float storage: never hand back non-finite features silently
            torch.cuda.synchronize(self.device)
            if self.lib.alink_resnet50_range_flag(self.h, 1):
                if self.dtype == "f16x2" and _retry:
                    self.calibrate(x, preprocessed, merge=True)
                    return self.embed_device(x, preprocessed, out=out, _retry=False)
                raise _abi.AlinkError("activations exceeded the float16 range in this network: use dtype='bf16' or 'f16x2'")
        return out

    def predict(self, X, batch_size=128, verbose=0, preprocessed=True):
        """Keras Model.predict on PREPROCESSED input (what RESNET50.process passes, code/siamese.py:216)."""
        torch = self.torch
        if isinstance(X, torch.Tensor):
            return self.embed_device(X.to("cuda:%d" % self.device), preprocessed)
        X = np.ascontiguousarray(np.asarray(X), dtype=np.float32)
        if len(X) == 0:
            return np.zeros((0, 2048), np.float32)
        return self.embed_device(torch.from_numpy(X).to("cuda:%d" % self.device), preprocessed).cpu().numpy()

    def profile(self, x):
        """One profiled forward on raw pixels: list of (op name, ms, flops)."""
        n = x.shape[0]
        out = self.torch.empty((n, 2048), dtype=self.torch.float32, device=x.device)
        ws, wsb = self._workspace(n)
        cap = 128
        ms, fl, k = (C.c_float * cap)(), (C.c_double * cap)(), C.c_int(cap)
        _abi.check(self.lib.alink_resnet50_profile(self.h, _abi.ptr(x), n, _abi.ptr(out), C.c_void_p(ws), wsb,
                                                   _abi.current_stream(self.device), ms, fl, C.byref(k)), "alink_resnet50_profile")
        return [(self.lib.alink_resnet50_op_name(self.h, i).decode(), ms[i], fl[i]) for i in range(k.value)]

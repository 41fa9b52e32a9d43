"""VGGFace16 — host-side owner of one alink_vgg16_t handle: the keras-vggface VGG-16 the reference
builds at code/siamese.py:187-193 (pool5 features, 25088-d at 224 x 224), with the `predict` slice of the
Keras Model API.  Weights: a Keras weight file (`rcmalli_vggface_tf_notop_vgg16.h5`, read with
hdf5_lite.py), a dict with Keras names, or synthetic when nothing is given (no network here)."""
import ctypes as C

import numpy as np

from . import _abi

BLOCKS = (2, 2, 3, 3, 3)
WIDTHS = (64, 128, 256, 512, 512)
MEAN_BGR = (93.5940, 104.7624, 129.1863)


def tensor_shapes():
    t, cin = {}, 3
    for b in range(5):
        for l in range(BLOCKS[b]):
            n = "conv%d_%d" % (b + 1, l + 1)
            t[n + "/kernel"] = (3, 3, cin, WIDTHS[b])
            t[n + "/bias"] = (WIDTHS[b],)
            cin = WIDTHS[b]
    return t


def synthetic_params(seed=1):
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in tensor_shapes().items():
        if name.endswith("/kernel"):
            v = rng.standard_normal(shape) * np.sqrt(2.0 / (9 * shape[2]))
        else:
            v = rng.standard_normal(shape) * 0.05
        p[name] = np.ascontiguousarray(v, dtype=np.float32)
    return p


def load_keras_h5(path):
    from . import hdf5_lite
    out = {}
    for lname, ws in hdf5_lite.load_keras_weights(path):
        for wname, arr in ws:
            out[lname + "/" + wname.rsplit("/", 1)[-1].split(":")[0]] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def save_keras_h5(path, params):
    from . import hdf5_lite
    layers = []
    for name in tensor_shapes():
        if name.endswith("/kernel"):
            l = name[:-len("/kernel")]
            layers.append((l, [(l + "/kernel:0", params[name]), (l + "/bias:0", params[l + "/bias"])]))
    hdf5_lite.save_keras_weights(path, layers)


class VGGFace16(object):
    def __init__(self, image_size=(224, 224), weights=None, dtype="bf16", device=None, max_batch=64, seed=1):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _abi.AlinkError("no ROCm device visible: a-link_amd computes only on the GPU (no CPU fallback)")
        device = _abi.resolve_device(device)          # None: the current torch device
        self.device = device
        self.lib = _abi.init(device)
        self.image_size = tuple(image_size)
        self.max_batch = int(max_batch)
        params = synthetic_params(seed) if weights is None else (load_keras_h5(weights) if isinstance(weights, str) else weights)
        with _abi.on_device(device):
            self.h = self.lib.alink_vgg16_create(int(image_size[0]), int(image_size[1]),
                                                 {"bf16": _abi.DT_BF16, "f16": _abi.DT_F16}[dtype])
        if not self.h:
            raise _abi.AlinkError("alink_vgg16_create: " + self.lib.alink_last_error().decode())
        name, cnt = C.c_char_p(), C.c_size_t()
        for i in range(self.lib.alink_vgg16_num_tensors(self.h)):
            _abi.check(self.lib.alink_vgg16_tensor_info(self.h, i, C.byref(name), C.byref(cnt)))
            key = name.value.decode()
            if key not in params:
                raise KeyError("weights are missing tensor %s" % key)
            a = np.ascontiguousarray(params[key], dtype=np.float32)
            _abi.check(self.lib.alink_vgg16_load(self.h, name.value, _abi.ptr(a), a.size), "load " + key)
        _abi.check(self.lib.alink_vgg16_finalize(self.h), "alink_vgg16_finalize")
        self.feature_size = self.lib.alink_vgg16_feature_size(self.h)
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.alink_vgg16_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _workspace(self, n):
        if self._ws is None or n > self._ws[1]:
            nbytes = self.lib.alink_vgg16_workspace_bytes(self.h, n)
            self._ws = (self.torch.empty(nbytes + 256, dtype=self.torch.uint8, device="cuda:%d" % self.device), n)
        t = self._ws[0]
        off = (-t.data_ptr()) % 256
        return t.data_ptr() + off, t.numel() - off

    def embed_device(self, x, preprocessed=False):
        torch = self.torch
        if x.ndim != 4 or tuple(x.shape[1:]) != self.image_size + (3,):
            raise ValueError("expected images of shape (N,%d,%d,3), got %s" % (self.image_size + (tuple(x.shape),)))
        x = x.to(torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty((n, self.feature_size), dtype=torch.float32, device=x.device)
        for i in range(0, n, self.max_batch):
            m = min(self.max_batch, n - i)
            ws, wsb = self._workspace(m)
            _abi.check(self.lib.alink_vgg16_embed(self.h, _abi.ptr(x[i:i + m]), m, 1 if preprocessed else 0,
                                                  _abi.ptr(out[i:i + m]), C.c_void_p(ws), wsb, _abi.current_stream(self.device)),
                       "alink_vgg16_embed")
        return out

    def predict(self, X, batch_size=32, verbose=0, preprocessed=True):
        torch = self.torch
        if isinstance(X, torch.Tensor):
            return self.embed_device(X.to("cuda:%d" % self.device), preprocessed)
        X = np.ascontiguousarray(np.asarray(X), dtype=np.float32)
        if len(X) == 0:
            return np.zeros((0, self.feature_size), np.float32)
        return self.embed_device(torch.from_numpy(X).to("cuda:%d" % self.device), preprocessed).cpu().numpy()

"""oracle/ir_resnet.py — unfused f32/f64 CPU forward of the insightface LResNet-E-IR backbone.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference: the symbol is not in
/root/reference — it is loaded from a downloaded checkpoint at reference code/face_model.py:34
(`mx.model.load_checkpoint`), sliced at `fc1_output` (code/face_model.py:35-36) and run at
code/face_model.py:90; the L2 normalisation is code/face_model.py:92 (sklearn.preprocessing.normalize).
Third-party source restated: insightface `src/symbols/fresnet.py` (mxnet, version unpinned —
requirements.txt does not list it), configuration version_input=1, version_output='E',
version_unit=3, act_type='prelu', BN eps 2e-5 — as summarised in SURVEY.md §8 row a5.

Layer by layer, no folding, no fusion — deliberately the opposite of the HIP path so that the parity
tests also check the BN folding / border-class bias / layout permutations done in backbone.hip.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 2e-5
_BN_INPUT_HOOK = None


def _t(params, name, dtype):
    return torch.as_tensor(np.asarray(params[name]), dtype=dtype)


def _bn(x, params, name, dtype, fix_gamma=False, eps=BN_EPS):
    """mx.sym.BatchNorm(use_global_stats / is_train=False): (x - mean) / sqrt(var + eps) * gamma + beta."""
    if _BN_INPUT_HOOK is not None:
        _BN_INPUT_HOOK(name, x, params)          # oracle/calibrate.py: set the statistics from this input
    g = _t(params, name + "_gamma", dtype)
    if fix_gamma:
        g = torch.ones_like(g)
    b = _t(params, name + "_beta", dtype)
    mu = _t(params, name + "_moving_mean", dtype)
    var = _t(params, name + "_moving_var", dtype)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mu.view(shape)) / torch.sqrt(var.view(shape) + eps) * g.view(shape) + b.view(shape)


def _prelu(x, params, name, dtype):
    """mx.sym.LeakyReLU(act_type='prelu'): per-channel slope `gamma`."""
    a = _t(params, name + "_gamma", dtype).view(1, -1, 1, 1)
    return torch.where(x > 0, x, x * a)


def infer_units(params):
    units = []
    for s in range(1, 5):
        u = 0
        while ("stage%d_unit%d_conv1_weight" % (s, u + 1)) in params:
            u += 1
        units.append(u)
    return units


def forward_raw(params, pixels_nchw, dtype=torch.float32, taps=None):
    """pixels_nchw: (N,3,H,W) float RGB 0..255 (what FaceModel.get_feature feeds MXNet after
    np.expand_dims, code/face_model.py:87-88).  Returns the un-normalised fc1 output (N, emb).
    `taps`, if a dict, receives intermediate NCHW tensors by layer name (for per-layer tests)."""
    x = pixels_nchw.to(dtype) if isinstance(pixels_nchw, torch.Tensor) else torch.as_tensor(np.asarray(pixels_nchw), dtype=dtype)
    units = infer_units(params)
    x = (x - 127.5) * 0.0078125
    x = F.conv2d(x, _t(params, "conv0_weight", dtype), stride=1, padding=1)
    x = _bn(x, params, "bn0", dtype)
    x = _prelu(x, params, "relu0", dtype)
    if taps is not None:
        taps["stem"] = x
    for s in range(4):
        for u in range(units[s]):
            p = "stage%d_unit%d" % (s + 1, u + 1)
            stride = 2 if u == 0 else 1
            y = _bn(x, params, p + "_bn1", dtype)
            y = F.conv2d(y, _t(params, p + "_conv1_weight", dtype), stride=1, padding=1)
            y = _bn(y, params, p + "_bn2", dtype)
            y = _prelu(y, params, p + "_relu1", dtype)
            if taps is not None:
                taps[p + "_conv1"] = y
            y = F.conv2d(y, _t(params, p + "_conv2_weight", dtype), stride=stride, padding=1)
            y = _bn(y, params, p + "_bn3", dtype)
            if u == 0:
                sc = F.conv2d(x, _t(params, p + "_conv1sc_weight", dtype), stride=stride, padding=0)
                sc = _bn(sc, params, p + "_sc", dtype)
            else:
                sc = x
            x = y + sc
            if taps is not None:
                taps[p] = x
    x = _bn(x, params, "bn1", dtype)
    # Dropout(p=0.4) is the identity at inference
    x = x.flatten(1)  # NCHW flatten: (C, H, W) order
    w = _t(params, "pre_fc1_weight", dtype)
    x = x @ w.t() + _t(params, "pre_fc1_bias", dtype)
    x = _bn(x, params, "fc1", dtype, fix_gamma=True)
    return x


def l2_normalize(e):
    """sklearn.preprocessing.normalize(X) (l2, axis=1): rows divided by their norm, zero norms -> 1
    (reference code/face_model.py:92)."""
    # sklearn's own arithmetic (sklearn/preprocessing/_data.py normalize -> utils.extmath.row_norms): squared norms by einsum IN
    # THE INPUT'S DTYPE, square root, zeros replaced by 1 — so that a float32 row whose squares underflow is left alone exactly
    # where sklearn leaves it alone (checked against sklearn itself: tests/test_oracle_sklearn_pins.py)
    e = np.asarray(e)
    n = np.sqrt(np.einsum("ij,ij->i", e, e))
    n[n == 0] = 1
    return e / n[:, None]


def embed(params, pixels_nhwc, dtype=torch.float32, batch=16):
    """siamese.ArcFace.process (reference code/siamese.py:232-234): per image get_input (HWC->CHW,
    code/face_model.py:83) then get_feature.  Batched here; the arithmetic per image is identical."""
    px = np.asarray(pixels_nhwc, dtype=np.float32)
    out = []
    with torch.no_grad():
        for i in range(0, len(px), batch):
            chw = np.transpose(px[i:i + batch], (0, 3, 1, 2))
            out.append(forward_raw(params, chw, dtype).to(torch.float32).numpy())
    return l2_normalize(np.concatenate(out, axis=0)).astype(np.float32)


def flops_per_image(units, widths=(64, 64, 128, 256, 512), size=112, emb=512):
    """Algorithmic conv+FC FLOPs (2*MAC) of one forward — SURVEY.md §8d: r100 = 24.179 GFLOP."""
    mac = size * size * 27 * widths[0]
    h = size
    for s in range(4):
        cin, c = widths[s], widths[s + 1]
        ho = h // 2
        mac += h * h * 9 * cin * c + ho * ho * 9 * c * c + ho * ho * cin * c
        mac += (units[s] - 1) * 2 * ho * ho * 9 * c * c
        h = ho
    mac += widths[4] * h * h * emb
    return 2 * mac

"""Backbone checkpoints: tensor name table (MXNet / insightface naming), synthetic initialisation and
the .npz side format.

The reference loads `model-r100-ii/model-symbol.json` + `model-0000.params` downloaded from Dropbox
(reference code/arcface_prepreq.sh:13-20, code/face_model.py:34).  Neither file is available offline,
so bench/tests use synthetic weights of the same architecture:
conv He-normal, BN gamma~U(.5,1.5) beta~N(0,.1) mean~N(0,.1) var~U(.5,1.5), PReLU 0.25, FC Xavier.
A real checkpoint converted to an .npz with the same tensor names loads through `load_npz`.
"""
import numpy as np

R100_UNITS = (3, 13, 30, 3)
R50_UNITS = (3, 4, 14, 3)
WIDTHS = (64, 64, 128, 256, 512)
ARCH_UNITS = {"r100": R100_UNITS, "r50": R50_UNITS, "r34": (3, 4, 6, 3), "r18": (2, 2, 2, 2)}


def tensor_shapes(units, widths=WIDTHS, size=(112, 112), emb=512):
    """Ordered {name: shape} of every tensor the LResNet-E-IR checkpoint holds (MXNet layouts)."""
    t = {}

    def bn(name, c):
        for s in ("_gamma", "_beta", "_moving_mean", "_moving_var"):
            t[name + s] = (c,)

    t["conv0_weight"] = (widths[0], 3, 3, 3)
    bn("bn0", widths[0])
    t["relu0_gamma"] = (widths[0],)
    h, w = size
    for s in range(4):
        c = widths[s + 1]
        for u in range(units[s]):
            p = "stage%d_unit%d" % (s + 1, u + 1)
            cin = widths[s] if u == 0 else c
            bn(p + "_bn1", cin)
            t[p + "_conv1_weight"] = (c, cin, 3, 3)
            bn(p + "_bn2", c)
            t[p + "_relu1_gamma"] = (c,)
            t[p + "_conv2_weight"] = (c, c, 3, 3)
            bn(p + "_bn3", c)
            if u == 0:
                t[p + "_conv1sc_weight"] = (c, cin, 1, 1)
                bn(p + "_sc", c)
        h, w = (h + 1) // 2, (w + 1) // 2
    bn("bn1", widths[4])
    t["pre_fc1_weight"] = (emb, widths[4] * h * w)
    t["pre_fc1_bias"] = (emb,)
    bn("fc1", emb)
    return t


def synthetic_ir_params(units=R100_UNITS, widths=WIDTHS, size=(112, 112), emb=512, seed=1, normalized=False):
    """normalized=True: same draws, then BatchNorm statistics set to match the activations (normalize_bn_statistics_)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in tensor_shapes(units, widths, size, emb).items():
        if name.endswith("_weight") and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif name == "pre_fc1_weight":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = rng.uniform(-lim, lim, shape)
        elif name == "pre_fc1_bias":
            v = rng.standard_normal(shape) * 0.01
        elif name.endswith("_gamma") and ("relu" in name):
            v = np.full(shape, 0.25)
        elif name.endswith("_gamma"):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("_beta") or name.endswith("_moving_mean"):
            v = rng.standard_normal(shape) * 0.1
        elif name.endswith("_moving_var"):
            v = rng.uniform(0.5, 1.5, shape)
        else:
            raise AssertionError(name)
        p[name] = np.ascontiguousarray(v, dtype=np.float32)
    if normalized:
        normalize_bn_statistics_(p, units, widths)
    return p


def _prelu_moments(mu, var, alpha):
    """mean and variance of PReLU_alpha(x), x ~ N(mu, var), per channel (closed form)."""
    from math import sqrt, pi
    sd = np.sqrt(np.maximum(var, 1e-30))
    t = mu / sd
    Phi = 0.5 * (1.0 + np.vectorize(__import__("math").erf)(t / sqrt(2.0)))
    phi = np.exp(-0.5 * t * t) / sqrt(2.0 * pi)
    e_relu = mu * Phi + sd * phi
    e_relu2 = (mu * mu + var) * Phi + mu * sd * phi
    m = alpha * mu + (1.0 - alpha) * e_relu
    m2 = alpha * alpha * (mu * mu + var) + (1.0 - alpha * alpha) * e_relu2
    return m, np.maximum(m2 - m * m, 1e-12)


def normalize_bn_statistics_(p, units, widths=WIDTHS):
    """In place: every BatchNorm's moving mean / variance := the mean / variance its input has when the pixels are
    independent uniform noise 0..255 (what bench.py and most tests feed), by propagating per-channel moments through
    the layers in closed form (pixels and channels treated as independent; borders ignored).  A trained checkpoint's
    statistics MATCH its activations; the SURVEY §8d draw (mean ~ N(0,.1), var ~ U(.5,1.5)) does not, so its
    activations grow ~1.5x per unit to ~1e8 at r100 — outside float16.  With these statistics they stay O(10) and the
    float16 storage type can be used (and measured) on synthetic weights.  Pure NumPy float64, deterministic; the
    data-dependent variant for arbitrary images is the test-side oracle/calibrate.py."""
    def conv_moments(w, m, v):
        w = np.asarray(w, np.float64)
        return w.sum(axis=(2, 3)) @ m, (w * w).sum(axis=(2, 3)) @ v

    def bn_set(name, m, v):
        p[name + "_moving_mean"] = m.astype(np.float32)
        p[name + "_moving_var"] = np.maximum(v, 1e-6).astype(np.float32)
        g = np.asarray(p[name + "_gamma"], np.float64)
        return np.asarray(p[name + "_beta"], np.float64), g * g          # output mean, variance

    m = np.full(3, (127.5 - 127.5) * 0.0078125)
    v = np.full(3, (256.0 ** 2 - 1.0) / 12.0 * 0.0078125 ** 2)
    m, v = conv_moments(p["conv0_weight"], m, v)
    m, v = bn_set("bn0", m, v)
    m, v = _prelu_moments(m, v, np.asarray(p["relu0_gamma"], np.float64))
    for s in range(4):
        for u in range(units[s]):
            q = "stage%d_unit%d" % (s + 1, u + 1)
            bm, bv = bn_set(q + "_bn1", m, v)
            bm, bv = conv_moments(p[q + "_conv1_weight"], bm, bv)
            bm, bv = bn_set(q + "_bn2", bm, bv)
            bm, bv = _prelu_moments(bm, bv, np.asarray(p[q + "_relu1_gamma"], np.float64))
            bm, bv = conv_moments(p[q + "_conv2_weight"], bm, bv)
            bm, bv = bn_set(q + "_bn3", bm, bv)
            if u == 0:
                sm, sv = conv_moments(p[q + "_conv1sc_weight"], m, v)
                sm, sv = bn_set(q + "_sc", sm, sv)
            else:
                sm, sv = m, v
            m, v = bm + sm, bv + sv
    m, v = bn_set("bn1", m, v)
    w = np.asarray(p["pre_fc1_weight"], np.float64)
    hw = w.shape[1] // len(m)
    fm = w @ np.repeat(m, hw) + np.asarray(p["pre_fc1_bias"], np.float64)      # C,H,W flatten: channel-major
    fv = (w * w) @ np.repeat(v, hw)
    p["fc1_moving_mean"] = fm.astype(np.float32)
    p["fc1_moving_var"] = np.maximum(fv, 1e-6).astype(np.float32)
    return p


def infer_units(params):
    units = []
    for s in range(1, 5):
        u = 0
        while ("stage%d_unit%d_conv1_weight" % (s, u + 1)) in params:
            u += 1
        units.append(u)
    return tuple(units)


def save_npz(path, params):
    np.savez(path, **params)


def load_npz(path):
    with np.load(path) as z:
        out = {}
        for k in z.files:
            # accept MXNet's "arg:" / "aux:" prefixes as written by mx.nd.save of a checkpoint dict
            name = k.split(":", 1)[1] if (k.startswith("arg:") or k.startswith("aux:")) else k
            out[name] = np.ascontiguousarray(z[k], dtype=np.float32)
        return out


def resolve_model_config(model_str, image_size=(112, 112)):
    """`args.model` of FaceModel ("prefix,epoch", reference code/face_model.py:29-33) ->
    (params dict, config dict with widths / bn_eps / emb).

    prefix forms:  synthetic:<arch>[:seed[:normalized]]   synthetic weights (bench / tests)
                   <path>                    an MXNet checkpoint <path>-symbol.json + <path>-%04d.params
                                             (what the reference loads, code/face_model.py:34), read by
                                             mxnet_format.py; else <path>-%04d.npz with the same names
    """
    import os
    vec = model_str.split(",")
    assert len(vec) == 2, "model must be 'prefix,epoch' (reference code/face_model.py:29-30)"
    prefix, epoch = vec[0], int(vec[1])
    cfg = {"widths": WIDTHS, "bn_eps": 2e-5, "emb": 512}
    if prefix.startswith("synthetic:"):
        parts = prefix.split(":")
        arch = parts[1]
        seed = int(parts[2]) if len(parts) > 2 else 1
        normalized = len(parts) > 3 and parts[3] == "normalized"
        return synthetic_ir_params(ARCH_UNITS[arch], size=image_size, seed=seed, normalized=normalized), cfg
    if os.path.exists("%s-symbol.json" % prefix) and os.path.exists("%s-%04d.params" % (prefix, epoch)):
        from . import mxnet_format as MX
        sym, arg, aux = MX.load_checkpoint(prefix, epoch)
        got = MX.ir_config_from_symbol(sym, "fc1_output")          # code/face_model.py:35-36
        params = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in list(arg.items()) + list(aux.items())}
        if infer_units(params) != got["units"]:
            raise ValueError("symbol describes units %s but the .params file holds %s" % (got["units"], infer_units(params)))
        cfg.update(widths=got["widths"], bn_eps=got["bn_eps"], emb=got["emb"])
        return params, cfg
    path = "%s-%04d.npz" % (prefix, epoch)
    return load_npz(path), cfg


def resolve_model(model_str, image_size=(112, 112)):
    return resolve_model_config(model_str, image_size)[0]

"""oracle/calibrate.py — synthetic LResNet-E-IR weights whose activations stay O(1) ("calibrated").

TEST INFRASTRUCTURE (see oracle/__init__.py).  The SURVEY §8d synthetic weights draw every BatchNorm's
moving mean / variance at random, so nothing normalises anything: activations grow by ~1.5x per unit and reach
~1e8 after r100's 49 units (float16 overflows; bf16 rounding accumulates on a scale no trained network has).
A trained checkpoint — what the reference loads at code/face_model.py:34 — has BatchNorm statistics that
MATCH its activations.  This module produces that property without training: the same random weights, then one
layer-by-layer pass of oracle/ir_resnet.py over a calibration batch in which every BatchNorm's moving mean /
variance are set to the per-channel statistics of the tensor arriving at it (data-dependent initialisation).
After it each BN output is ~N(beta, gamma^2) on the calibration distribution and the residual stream grows
only like sqrt(units).

PARITY UNPINNED like the rest of the CNN oracle: this changes the weights the parity tests run on, not what
pins the arithmetic.
"""
import numpy as np
import torch

from . import ir_resnet


def calibration_pixels(n, size, seed=1234):
    """Smooth-ish random 'images' (low-frequency field + pixel noise, 0..255), so that neighbouring pixels
    correlate like a photograph's do; uniform noise alone has no spatial structure for the 3x3 kernels."""
    rng = np.random.default_rng(seed)
    h, w = size
    coarse = rng.uniform(0, 255, (n, (h + 7) // 8 + 1, (w + 7) // 8 + 1, 3)).astype(np.float32)
    t = torch.from_numpy(coarse).permute(0, 3, 1, 2)
    up = torch.nn.functional.interpolate(t, size=(h, w), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).numpy()
    px = 0.75 * up + 0.25 * rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    return np.clip(np.rint(px), 0, 255).astype(np.float32)


def calibrate_(params, pixels_nhwc, var_floor=1e-3, batch=None):
    """In place: every *_moving_mean / *_moving_var := statistics of that BatchNorm's input on `pixels`."""
    def hook(name, x, p):
        dims = [0] + list(range(2, x.dim()))
        mu = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False).clamp_min(var_floor)
        p[name + "_moving_mean"] = mu.to(torch.float32).numpy().copy()
        p[name + "_moving_var"] = var.to(torch.float32).numpy().copy()

    chw = np.transpose(np.asarray(pixels_nhwc, np.float32), (0, 3, 1, 2))
    ir_resnet._BN_INPUT_HOOK = hook
    try:
        with torch.no_grad():
            ir_resnet.forward_raw(params, chw, torch.float32)
    finally:
        ir_resnet._BN_INPUT_HOOK = None
    return params


def calibrated_ir_params(units, size=(112, 112), seed=1, n_cal=16, widths=(64, 64, 128, 256, 512), emb=512):
    import a_link_amd  # noqa: F401  (package alias)
    from a_link_amd import weights as W
    params = W.synthetic_ir_params(units, widths=widths, size=size, emb=emb, seed=seed)
    return calibrate_(params, calibration_pixels(n_cal, size, seed=1000 + seed))


def activation_range(params, pixels_nhwc):
    """max |activation| over every unit output + the stem, for reporting."""
    taps = {}
    with torch.no_grad():
        ir_resnet.forward_raw(params, np.transpose(np.asarray(pixels_nhwc, np.float32), (0, 3, 1, 2)), torch.float32, taps)
    return max(float(v.abs().max()) for v in taps.values())

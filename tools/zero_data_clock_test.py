#!/usr/bin/env python3
"""Is the embedding rate held down by the clock the chip keeps under MFMA load on real data (MI355X_MICROARCH.md, DVFS
give-back)?  Same launches, same kernels: random weights + random pixels against all-zero weights (every activation and
every MFMA operand zero).  A large gap = the plateau every launch shape lands on is the chip's power / clock limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import weights as W
from a_link_amd.backbone import IRBackbone


def rate(params, x, reps=12):
    bb = IRBackbone(params, max_batch=292, streams=4)
    out = torch.empty((x.shape[0], 512), device="cuda")
    for _ in range(3):
        bb.embed_device(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        bb.embed_device(x, out)
    torch.cuda.synchronize()
    return reps * x.shape[0] / (time.perf_counter() - t0)


def main():
    p = W.synthetic_ir_params(W.R100_UNITS, seed=1)
    x = torch.randint(0, 256, (1168, 112, 112, 3), dtype=torch.uint8).float().cuda()
    print("random weights, random pixels: %.0f embeddings/s" % rate(p, x))
    pz = {k: (np.zeros_like(v) if k.endswith("_weight") else v) for k, v in p.items()}
    print("zero conv / FC weights (all-zero MFMA operands on one side, constant activations): %.0f embeddings/s" % rate(pz, x))
    pzz = {k: (np.zeros_like(v) if (k.endswith("_weight") or k.endswith("_beta") or k.endswith("_bias") or k.endswith("_moving_mean")) else v)
           for k, v in p.items()}
    print("zero weights and zero biases (every activation exactly zero): %.0f embeddings/s" % rate(pzz, torch.zeros_like(x)))
    print("random again: %.0f embeddings/s" % rate(p, x))


def soak(mode, seconds=12.0):
    p = W.synthetic_ir_params(W.R100_UNITS, seed=1)
    x = torch.randint(0, 256, (1168, 112, 112, 3), dtype=torch.uint8).float().cuda()
    if mode == "zeros":
        p = {k: (np.zeros_like(v) if (k.endswith("_weight") or k.endswith("_beta") or k.endswith("_bias") or k.endswith("_moving_mean")) else v)
             for k, v in p.items()}
        x = torch.zeros_like(x)
    bb = IRBackbone(p, max_batch=292, streams=4)
    out = torch.empty((1168, 512), device="cuda")
    for _ in range(3):
        bb.embed_device(x, out)
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            bb.embed_device(x, out)
        torch.cuda.synchronize()
        n += 20
    print("%s: %.0f embeddings/s over %.1f s" % (mode, n * 1168 / (time.perf_counter() - t0), time.perf_counter() - t0))


if __name__ == "__main__" and "--soak" in sys.argv:
    soak(sys.argv[sys.argv.index("--soak") + 1])
elif __name__ == "__main__" and "--dominant" not in sys.argv:
    main()


def dominant_launch_us(params, x):
    """mean duration of the stage-3 3x3 launches of one forward, each launch alone (alink_embed_profile)."""
    bb = IRBackbone(params, max_batch=292)
    for _ in range(2):
        bb.embed_device(x[:292])
    prof = bb.profile(x[:292])
    conv = [(ms, f) for k, ms, f in prof if k == 1]
    groups = {}
    for ms, f in conv:
        groups.setdefault(round(f / 1e6), []).append(ms)
    key = max(groups, key=lambda k: sum(groups[k]))             # the shape with the most time: 14x14, 256 -> 256
    return 1e3 * float(np.mean(groups[key])), key * 1e6, len(groups[key])


if __name__ == "__main__" and "--dominant" in sys.argv:
    p = W.synthetic_ir_params(W.R100_UNITS, seed=1)
    x = torch.randint(0, 256, (292, 112, 112, 3), dtype=torch.uint8).float().cuda()
    us, fl, n = dominant_launch_us(p, x)
    print("dominant launch on data:  %.1f us x %d launches -> %.0f TFLOP/s = %.1f %% of 2516.6" % (us, n, fl / us / 1e6, 100 * fl / us / 1e6 / 2516.6))
    pzz = {k: (np.zeros_like(v) if (k.endswith("_weight") or k.endswith("_beta") or k.endswith("_bias") or k.endswith("_moving_mean")) else v)
           for k, v in p.items()}
    us, fl, n = dominant_launch_us(pzz, torch.zeros_like(x))
    print("dominant launch on zeros: %.1f us x %d launches -> %.0f TFLOP/s = %.1f %% of 2516.6" % (us, n, fl / us / 1e6, 100 * fl / us / 1e6 / 2516.6))

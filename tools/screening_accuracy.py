#!/usr/bin/env python3
"""Accuracy and rate of the screening forms against the exact mode (split precision, three products) on the same weights and
pixels: plain bf16, plain f16 (where the activations fit), and the ONE-product form of the split-precision handle
(alink_backbone_set_products(1): f16 operands under the handle's scales, residual stream carried as hi + lo)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import a_link_amd  # noqa: F401
import bench
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone


def rate(fn, x, reps=3):
    fn(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn(x)
    torch.cuda.synchronize()
    return reps * x.shape[0] / (time.perf_counter() - t)


def main():
    pool, _ = bench._identity_pool(2336, 7)
    for arch, normalized in (("r50", True), ("r100", True), ("r100", False)):
        params = W.synthetic_ir_params(W.ARCH_UNITS[arch], seed=1, normalized=normalized)
        ex = IRBackbone(params, dtype="f16x2")
        ex.calibrate(pool[:292])
        ref = ex.embed_device(pool).double()
        forms = [("f16x2 (exact, 3 products)", ex.embed_device), ("f16x2/1 (one product, same handle)", ex.screening_view().embed_device)]
        for dt in ("f16", "bf16"):
            try:
                b = IRBackbone(params, dtype=dt)
                b.embed_device(pool[:292])
                forms.append((dt, b.embed_device))
            except _abi.AlinkError as e:
                print("%s %s %s: %s" % (arch, "normalized" if normalized else "survey", dt, str(e)[:60]))
        print("[%s, %s weights, %d identity-pool images]" % (arch, "normalized" if normalized else "survey", pool.shape[0]))
        for name, fn in forms:
            e = fn(pool).double()
            d = (e - ref).norm(dim=1)
            cos = 1.0 - (e * ref).sum(1)
            print("   %-36s |de| max %.2e mean %.2e   1-cos max %.2e   %8.0f emb/s" % (name, d.max(), d.mean(), cos.max(), rate(fn, pool)))
        del ex, forms


if __name__ == "__main__":
    main()

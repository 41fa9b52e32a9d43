#!/usr/bin/env python3
"""Up to how many output pixels per launch does the latency form (conv3x3_lat.hip) beat the tile kernels?  Forward time of n images
for a few limits, IR-100, per dtype; every cell is checked bit-equal against the tile kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone
lib = _abi.load()
params = W.synthetic_ir_params(W.ARCH_UNITS["r100"], seed=1, normalized=True)
limits = [int(a) for a in sys.argv[1:]] or [0, 784, 1600]
for dt in ("bf16", "f16x2"):
    bb = IRBackbone(params, dtype=dt, max_batch=292)
    for n in (1, 2, 4, 8, 16, 32, 64):
        x = torch.randint(0, 256, (n, 112, 112, 3), dtype=torch.uint8).cuda()
        res, outs = [], []
        for lim in limits:
            lib.alink_debug_set_latency_form(lim)
            for _ in range(5):
                o = bb.embed_device(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(30):
                o = bb.embed_device(x)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t) / 30 * 1e3)
            outs.append(o.clone())
        print("%s n=%2d: limits %s -> %s ms | bit-equal %s" % (dt, n, limits, " ".join("%.3f" % r for r in res), all(torch.equal(outs[0], o) for o in outs[1:])))
lib.alink_debug_set_latency_form(1600)

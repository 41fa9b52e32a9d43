#!/usr/bin/env python3
"""A/B of the latency form of the 3x3 stride-1 convolution (conv3x3_lat.hip: one wave per 32 x 32 output block, operands straight
from L2, no LDS / barrier) against the tile kernels on small batches: bit-equality of the embeddings and forward time, IR-100."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone
lib = _abi.load()
params = W.synthetic_ir_params(W.ARCH_UNITS["r100"], seed=1, normalized=True)
limits = [784]
forms = [int(a) for a in sys.argv[1:]] or [0, 4, 2]      # 0: 16x16 blocks, 36 sub-steps in flight; 4: the same with 18; 2: 32x32 blocks
for dt in ("bf16", "f16x2"):
    bb = IRBackbone(params, dtype=dt, max_batch=292)
    for n in (1, 2, 4, 8, 16):
        x = torch.randint(0, 256, (n, 112, 112, 3), dtype=torch.uint8).cuda()
        res, outs = [], []
        for form in [-1] + forms:
            lib.alink_debug_set_latency_form(0 if form < 0 else limits[0])
            lib.alink_debug_set_latency_tiles(max(form, 0))
            for _ in range(5):
                o = bb.embed_device(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(30):
                o = bb.embed_device(x)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t) / 30 * 1e3)
            outs.append(o.clone())
        print("%s n=%2d: tile kernels %.3f ms | latency forms %s: %s ms | bit-equal %s" % (
            dt, n, res[0], forms, " ".join("%.3f" % r for r in res[1:]), all(torch.equal(outs[0], o) for o in outs[1:])))
lib.alink_debug_set_latency_form(784)
lib.alink_debug_set_latency_tiles(0)

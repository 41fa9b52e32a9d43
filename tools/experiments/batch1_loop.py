#!/usr/bin/env python3
"""One image at a time, 100 forwards per dtype (for rocprofv3 --kernel-trace: the latency-form kernels' own durations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import a_link_amd  # noqa
from a_link_amd import weights as W
from a_link_amd.backbone import IRBackbone
params = W.synthetic_ir_params(W.ARCH_UNITS["r100"], seed=1, normalized=True)
x = torch.randint(0, 256, (1, 112, 112, 3), dtype=torch.uint8).cuda()
for dt in (sys.argv[1:] or ["f16x2"]):
    bb = IRBackbone(params, dtype=dt, max_batch=292, lazy_range_check=True)
    for _ in range(100):
        bb.embed_device(x)
    torch.cuda.synchronize()

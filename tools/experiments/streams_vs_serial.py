#!/usr/bin/env python3
"""2,048 images through a shallow net (units 1,1,1,1) on 4 streams against the one-stream result, REPS times: rows that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
if os.environ.get("ALINK_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["ALINK_LIB"])
from a_link_amd.backbone import IRBackbone
lib = _abi.load()
p = W.synthetic_ir_params((1, 1, 1, 1), seed=1, normalized=True)
x = torch.randint(0, 256, (2048, 112, 112, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
reps = int(os.environ.get("REPS", "60"))
for dtype in sys.argv[1:] or ["f16", "bf16"]:
    ref = IRBackbone(p, dtype=dtype, max_batch=292, streams=1, shards_per_call=1, lazy_range_check=True).embed_device(x).clone()
    bb = IRBackbone(p, dtype=dtype, max_batch=292, streams=4, lazy_range_check=True)
    tot, rows = 0, []
    for rep in range(reps):
        got = bb.embed_device(x).clone()
        torch.cuda.synchronize()
        bad = (got != ref).any(1).nonzero().flatten().tolist()
        tot += len(bad); rows += bad[:4]
    print(os.environ.get("ALINK_LIB", "library"), dtype, "4 streams against 1: differing rows over %d reps:" % reps, tot, rows[:12], flush=True)

#!/usr/bin/env python3
"""Where a workgroup's cycles go in front_c64_kernel (diagnostic STAMP build: wave 0 sums s_memtime differences; no product
call executes a stamp): barrier wait, stem rows, conv1 + epilogue, per pass."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone
lib = _abi.load()
lib.alink_debug_set_stamps.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 292
bb = IRBackbone(W.synthetic_ir_params((1, 1, 1, 1), seed=1), dtype="bf16", max_batch=n)
bb._set_shards(1) if hasattr(bb, "_set_shards") else None
x = torch.randint(0, 256, (n, 112, 112, 3), dtype=torch.uint8).cuda()
for _ in range(3):
    bb.profile(x)
st = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
lib.alink_debug_set_stamps(C.c_void_p(st.data_ptr()))
lib.alink_debug_set_profile_reps(1)
bb.profile(x)
lib.alink_debug_set_stamps(None)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
s = s[s[:, 0] != 0]
npass = s[:, 5]
print("%d workgroups, passes per workgroup %.1f; median cycles: total %.0f | prologues %.0f | per pass: barrier wait %.0f, stem rows %.0f, conv1 + epilogue %.0f"
      % (len(s), np.median(npass), np.median(s[:, 0]), np.median(s[:, 4]), np.median(s[:, 1] / npass), np.median(s[:, 2] / npass), np.median(s[:, 3] / npass)))
print("  p10/p90 of total: %.0f / %.0f" % (np.percentile(s[:, 0], 10), np.percentile(s[:, 0], 90)))

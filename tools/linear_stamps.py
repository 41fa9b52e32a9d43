#!/usr/bin/env python3
"""Per-workgroup s_memtime timeline of one conv3x3_linear launch (the stamps are skipped by every product call:
ConvParams::stamps is null there): where a workgroup's cycles go — prologue, main loop (with the time inside the
single-buffered input refills), epilogue — and how the workgroups of a launch line up."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi
lib = _abi.init(0)
lib.alink_debug_set_stamps.argtypes = [C.c_void_p]
shapes = {"s3": (292, 14, 14, 256, 256), "s2": (292, 28, 28, 128, 128), "s4": (292, 7, 7, 512, 512), "s1": (292, 56, 56, 64, 64)}
for name in sys.argv[1:] or ["s3", "s2", "s4", "s1"]:
    N, H, W, Ci, Co = shapes[name]
    x = torch.randn(N, H, W, Ci, device="cuda").bfloat16()
    w = (torch.randn(Co, 3, 3, Ci, device="cuda") * 0.03).bfloat16()
    b = torch.zeros(9, Co, device="cuda")
    r = torch.randn(N, H, W, Co, device="cuda").bfloat16()
    out = torch.empty(N, H, W, Co, device="cuda", dtype=torch.bfloat16)
    nwg = 16384
    st = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    for rep in range(3):
        lib.alink_debug_set_stamps(C.c_void_p(st.data_ptr()) if rep == 2 else None)
        _abi.check(lib.alink_conv_nhwc(0, _abi.ptr(x), _abi.ptr(w), _abi.ptr(b), None, _abi.ptr(r), _abi.ptr(out),
                                       N, H, W, Ci, Co, 3, 1, 1, 1, 0, None))
    lib.alink_debug_set_stamps(None)
    s = st.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    d = np.diff(s[:, :4], axis=1).astype(np.float64)
    span = s[:, 3].max() - s[:, 0].min()
    print("%s: %d workgroups; median cycles: prologue %.0f | loop %.0f (of which %d refills %.0f) | epilogue %.0f | total %.0f ; launch span %.0f"
          % (name, len(s), np.median(d[:, 0]), np.median(d[:, 1]), int(np.median(s[:, 5])), np.median(s[:, 4]), np.median(d[:, 2]),
             np.median(d.sum(1)), span))
    print("    epilogue split: loop end -> residual loads back %.0f | -> last store issued %.0f | -> stores drained %.0f"
          % (np.median(s[:, 6] - s[:, 2]), np.median(s[:, 7] - s[:, 6]), np.median(s[:, 3] - s[:, 7])))
    print("    p10/p90: prologue %.0f/%.0f loop %.0f/%.0f epilogue %.0f/%.0f ; start spread %.0f ; end spread %.0f"
          % (np.percentile(d[:, 0], 10), np.percentile(d[:, 0], 90), np.percentile(d[:, 1], 10), np.percentile(d[:, 1], 90),
             np.percentile(d[:, 2], 10), np.percentile(d[:, 2], 90), s[:, 0].max() - s[:, 0].min(), s[:, 3].max() - s[:, 3].min()))

#!/usr/bin/env python3
"""Per-workgroup s_memtime timeline of one conv3x3_direct launch (diagnostic build path)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi
lib = _abi.init(0)
lib.alink_debug_set_stamps.argtypes = [C.c_void_p]
lib.alink_debug_set_linear(0)            # the stamps live in the row-aligned kernel (conv3x3_direct.hip)
shapes = {"s2": (256, 28, 28, 128, 128), "s3": (256, 14, 14, 256, 256), "s1": (256, 56, 56, 64, 64), "s0": (256, 112, 112, 64, 64)}
for name in sys.argv[1:] or ["s2", "s3", "s1"]:
    N, H, W, Ci, Co = shapes[name]
    x = torch.randn(N, H, W, Ci, device="cuda").bfloat16()
    w = (torch.randn(Co, 3, 3, Ci, device="cuda") * 0.03).bfloat16()
    b = torch.zeros(9, Co, device="cuda")
    out = torch.empty(N, H, W, Co, device="cuda", dtype=torch.bfloat16)
    nwg = 16384
    st = torch.zeros(nwg * 4, dtype=torch.int64, device="cuda")
    for rep in range(3):
        lib.alink_debug_set_stamps(C.c_void_p(st.data_ptr()) if rep == 2 else None)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0.record()
        _abi.check(lib.alink_conv_nhwc(0, _abi.ptr(x), _abi.ptr(w), _abi.ptr(b), None, _abi.ptr(out), _abi.ptr(out),
                                       N, H, W, Ci, Co, 3, 1, 1, 1, -1, None))
        t1.record(); torch.cuda.synchronize()
    lib.alink_debug_set_stamps(None)
    s = st.cpu().numpy().reshape(-1, 4)
    s = s[s[:, 0] != 0]
    d = np.diff(s, axis=1).astype(np.float64)
    t_all = (s[:, 3].max() - s[:, 0].min())
    print("%s: %d WGs; cycles median: prologue %.0f  loop %.0f  epilogue %.0f  | total/WG %.0f ; kernel span %.0f cyc"
          % (name, len(s), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(d.sum(1)), t_all))
    print("    event time %.1f us -> %.0f MHz if the span is the kernel" % (t0.elapsed_time(t1) * 1e3, t_all / (t0.elapsed_time(t1) * 1e3)))
    print("    p10/p90 prologue %.0f/%.0f loop %.0f/%.0f epi %.0f/%.0f ; start spread %.0f"
          % (np.percentile(d[:, 0], 10), np.percentile(d[:, 0], 90), np.percentile(d[:, 1], 10), np.percentile(d[:, 1], 90),
             np.percentile(d[:, 2], 10), np.percentile(d[:, 2], 90), s[:, 0].max() - s[:, 0].min()))

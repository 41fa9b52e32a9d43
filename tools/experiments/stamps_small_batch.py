import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import a_link_amd
from a_link_amd import _abi
lib = _abi.init(0)
lib.alink_debug_set_stamps.argtypes = [C.c_void_p]
for name, (N, H, W, Ci, Co) in {"s3 b1": (1, 14, 14, 256, 256), "s2 b1": (1, 28, 28, 128, 128), "s4 b1": (1, 7, 7, 512, 512), "s3 b16": (16, 14, 14, 256, 256)}.items():
    x = torch.randn(N, H, W, Ci, device="cuda").bfloat16()
    w = (torch.randn(Co, 3, 3, Ci, device="cuda") * 0.03).bfloat16()
    b = torch.zeros(9, Co, device="cuda")
    r = torch.randn(N, H, W, Co, device="cuda").bfloat16()
    out = torch.empty(N, H, W, Co, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(16384 * 8, dtype=torch.int64, device="cuda")
    for fine in (1, 0):
        for rep in range(3):
            lib.alink_debug_set_stamps(C.c_void_p(st.data_ptr()) if rep == 2 else None)
            _abi.check(lib.alink_conv_nhwc(0, _abi.ptr(x), _abi.ptr(w), _abi.ptr(b), None, _abi.ptr(r), _abi.ptr(out), N, H, W, Ci, Co, 3, 1, 1, 1, fine, None))
        lib.alink_debug_set_stamps(None)
        torch.cuda.synchronize()
        s = st.cpu().numpy().reshape(-1, 8); s = s[s[:, 0] != 0]
        d = np.diff(s[:, :4], axis=1).astype(np.float64)
        nk = 9 * Ci // 64
        print("%s fine=%d: %d workgroups; prologue %.0f | loop %.0f = %d K-steps x %.0f (refills %d: %.0f) | epilogue %.0f | total %.0f" % (name, fine, len(s), np.median(d[:, 0]), np.median(d[:, 1]), nk, (np.median(d[:, 1]) - np.median(s[:, 4])) / nk, int(np.median(s[:, 5])), np.median(s[:, 4]), np.median(d[:, 2]), np.median(d.sum(1))))
        st.zero_()

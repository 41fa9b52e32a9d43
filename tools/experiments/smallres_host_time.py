"""SmallRes train_on_batch: how much of a step is the HOST enqueueing it?  (time until the C call returns vs until the stream is idle)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
import a_link_amd
from a_link_amd import _abi
from a_link_amd.smallres import SmallResNet
rs = np.random.RandomState(0)
L = torch.from_numpy(((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
R = torch.from_numpy(((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
y = torch.from_numpy(np.eye(2, dtype=np.float32)[rs.randint(0, 2, 16)]).cuda()
srn = SmallResNet((32, 32, 3), 2048, lr=0.1, seed=1)
for _ in range(10):
    srn.train_on_batch([L, R], y)
e1, e2 = srn.mask_sizes
md = torch.empty(2 * 16 * (e1 + e2), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
out = {}
for overlap in (1, 0):
    srn.lib.alink_debug_set_smallres_overlap(overlap)
    enq, tot = [], []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _abi.check(srn.lib.alink_smallres_train_step(srn.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(y), None, 16, 0, _abi.ptr(md), 0.0, 1,
                                                     C.c_void_p(srn._metrics_host.data_ptr()), C.c_void_p(st.cuda_stream)))
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    out["overlap_%d" % overlap] = {"enqueue_ms": 1e3 * float(np.median(enq)), "until_idle_ms": 1e3 * float(np.median(tot))}
print(json.dumps(out))

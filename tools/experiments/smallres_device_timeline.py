"""SmallRes train step under `rocprofv3 --kernel-trace` WITHOUT the profiler's host overhead in the picture: every step is
enqueued behind a ~1 ms spin kernel, so the host has finished enqueueing before the device starts and the trace shows the step as
the device alone runs it (kernel starts / ends on both streams).   bash tools/trace.sh <name> tools/experiments/smallres_device_timeline.py [steps] [one_update] [mini]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
import a_link_amd  # noqa: F401
from a_link_amd import _abi
from a_link_amd.smallres import SmallResNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
one = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mini = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rs = np.random.RandomState(0)
L = torch.from_numpy(((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
R = torch.from_numpy(((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
y = torch.from_numpy(np.eye(2, dtype=np.float32)[rs.randint(0, 2, 16)]).cuda()
srn = SmallResNet((32, 32, 3), 2048, lr=0.1, seed=1)
for _ in range(5):
    srn.train_on_batch([L, R], y)
e1, e2 = srn.mask_sizes
md = torch.empty(2 * 16 * (e1 + e2), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
srn.lib.alink_debug_set_smallres_one_update(one)
srn.lib.alink_debug_set_mini_step(mini)
for _ in range(steps):
    torch.cuda.synchronize()
    torch.cuda._sleep(2400000)
    _abi.check(srn.lib.alink_smallres_train_step(srn.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(y), None, 16, 0, _abi.ptr(md), 0.0, 1,
                                                 C.c_void_p(srn._metrics_host.data_ptr()), C.c_void_p(st.cuda_stream)))
torch.cuda.synchronize()

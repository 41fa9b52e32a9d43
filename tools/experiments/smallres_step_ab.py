"""SmallRes train step (16 pairs, 32 x 32, 2048 features): A/B of the round-6 launch savings ON ONE BOX, interleaved —
    one_update   the tower's update + the head's as one launch (alink_debug_set_smallres_one_update)
    mini_step    the head's train step + input gradients as three launches (alink_debug_set_mini_step)
per setting: time until the C call returns (host enqueue) and until the stream is idle, medians over `reps` steps."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
import a_link_amd  # noqa: F401
from a_link_amd import _abi
from a_link_amd.smallres import SmallResNet

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
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
lib = srn.lib
settings = [("default", 1, 1), ("two_update_launches", 0, 1), ("generic_head_chain", 1, 0), ("both_off", 0, 0)]
acc = {k: ([], []) for k, _, _ in settings}
for rnd in range(4):                       # interleaved rounds: drift of the box shows up in every setting alike
    for name, one, mini in settings:
        lib.alink_debug_set_smallres_one_update(one)
        lib.alink_debug_set_mini_step(mini)
        for i in range(reps // 4 + 5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _abi.check(lib.alink_smallres_train_step(srn.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(y), None, 16, 0, _abi.ptr(md), 0.0, 1,
                                                     C.c_void_p(srn._metrics_host.data_ptr()), C.c_void_p(st.cuda_stream)))
            t1 = time.perf_counter()
            st.synchronize()
            t2 = time.perf_counter()
            if i >= 5:
                acc[name][0].append(t1 - t0)
                acc[name][1].append(t2 - t0)
lib.alink_debug_set_smallres_one_update(1)
lib.alink_debug_set_mini_step(1)
print(json.dumps({k: {"enqueue_ms": round(1e3 * float(np.median(v[0])), 4), "until_idle_ms": round(1e3 * float(np.median(v[1])), 4)}
                  for k, v in acc.items()}))

"""A/B: SmallRes train_on_batch as plain launches on torch's stream / plain launches on the model's own stream with staging /
one captured graph."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd
from a_link_amd import _abi
from a_link_amd.smallres import SmallResNet
rs = np.random.RandomState(0)
L = torch.from_numpy(((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
R = torch.from_numpy(((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
y = torch.from_numpy(np.eye(2, dtype=np.float32)[rs.randint(0, 2, 16)]).cuda()
out = {}
for name, use_graph, c_graph in (("plain_current_stream", False, 0), ("own_stream_no_graph", True, 0), ("own_stream_graph", True, 1)):
    srn = SmallResNet((32, 32, 3), 2048, lr=0.1, seed=1)
    srn.use_graph = use_graph                       # the model's own stream + staging buffers
    _abi.check(srn.lib.alink_smallres_set_graph(srn.h, c_graph))
    np.random.seed(0)
    for _ in range(10):
        srn.train_on_batch([L, R], y)
    ts = []
    for _ in range(200):
        torch.cuda.synchronize()
        t = time.perf_counter()
        srn.train_on_batch([L, R], y)
        ts.append(time.perf_counter() - t)
    out[name] = 1e3 * float(np.median(ts))
print(json.dumps(out))

import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from a_link_amd.smallres import SmallResNet
srn = SmallResNet((32, 32, 3), 2048, lr=0.1, seed=1)
rs_ = np.random.RandomState(0)
sL = ((rs_.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
sR = ((rs_.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
sy = np.eye(2, dtype=np.float32)[rs_.randint(0, 2, 16)]
np.random.seed(0)
for _ in range(10):
    srn.train_on_batch([sL, sR], sy)
out = {}
for name, args in (("host", ([sL, sR], sy)), ("device", ([torch.from_numpy(sL).cuda(), torch.from_numpy(sR).cuda()], torch.from_numpy(sy).cuda()))):
    ts = []
    for _ in range(300):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        srn.train_on_batch(*args)
        ts.append(time.perf_counter() - t1)
    out[name] = 1e3 * float(np.median(ts))
print(json.dumps(out))

import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd
from a_link_amd import siamese
for res, feat in ((32, 2048), (48, 2048)):
    m = siamese.SmallRes((res, res, 3), (feat,), "x", 0.1, seed=1)
    rng = np.random.RandomState(0)
    L = rng.randint(0, 256, (4096, res, res, 3)).astype(np.float32); R = rng.randint(0, 256, (4096, res, res, 3)).astype(np.float32)
    Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m.predict([Ld, Rd]); torch.cuda.synchronize()
    t = time.perf_counter(); 
    for _ in range(3): m.predict([Ld, Rd])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    macs = {32: 24.27e6, 48: 60.61e6}[res]
    print("SmallRes-%d predict 4096 pairs: %.2f ms  (%.0f pairs/s, %.2f TFLOP/s)" % (res, dt * 1e3, 4096 / dt, 2 * 2 * macs * 4096 / dt / 1e12))
    y = np.eye(2, dtype=np.float32)[rng.randint(0, 2, 16)]
    xs = [(L[:16] - 128) / 128, (R[:16] - 128) / 128]
    for _ in range(5): m.siamese_net.train_on_batch(xs, y)
    ts = []
    for _ in range(50):
        torch.cuda.synchronize(); t = time.perf_counter(); m.siamese_net.train_on_batch(xs, y); ts.append(time.perf_counter() - t)
    print("SmallRes-%d train_on_batch(16): %.3f ms" % (res, 1e3 * np.median(ts)))

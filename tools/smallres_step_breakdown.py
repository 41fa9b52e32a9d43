import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import a_link_amd
from a_link_amd.smallres import SmallResNet
net = SmallResNet((32, 32, 3), 2048, lr=0.1, seed=1)
rs = np.random.RandomState(0)
L = ((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
R = ((rs.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
y = np.eye(2, dtype=np.float32)[rs.randint(0, 2, 16)]
def med(fn, n=50):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
print("train_on_batch (host arrays, masks drawn)      %.3f ms" % med(lambda: net.train_on_batch([L, R], y)))
m = net.draw_masks(16)
print("draw_masks alone                               %.3f ms" % med(lambda: net.draw_masks(16)))
print("train_on_batch (host arrays, masks given)      %.3f ms" % med(lambda: net.train_on_batch([L, R], y, masks=m)))
Ld, Rd, yd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), torch.from_numpy(y).cuda()
md = torch.from_numpy(m).cuda()
net.training_dropout = False
print("train_on_batch (device arrays, no dropout)     %.3f ms" % med(lambda: net.train_on_batch([Ld, Rd], yd)))

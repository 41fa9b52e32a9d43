"""one-rank RCCL: time of the sharded fine-tune step and of its parts (debug of a 0.085 -> 0.47 ms reading)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import a_link_amd  # noqa
from a_link_amd import distributed as D
from a_link_amd.head import DenseHead
rng = np.random.RandomState(0)
L = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda(); R = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda()
y = np.zeros((16, 2), np.float32); y[np.arange(16), rng.randint(0, 2, 16)] = 1; yd = torch.from_numpy(y).cuda()
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
hd = DenseHead(512, lr=0.1, seed=0, device=0)
print("sharded allreduce  %.4f ms" % t(lambda: D.dp_train_on_batch(hd, [L, R], yd, mode="sharded", exchange="allreduce")))
print("sharded gather     %.4f ms" % t(lambda: D.dp_train_on_batch(hd, [L, R], yd, mode="sharded")))
print("sharded allreduce  %.4f ms" % t(lambda: D.dp_train_on_batch(hd, [L, R], yd, mode="sharded", exchange="allreduce")))
print("sharded gather     %.4f ms" % t(lambda: D.dp_train_on_batch(hd, [L, R], yd, mode="sharded")))
print("replicated         %.4f ms" % t(lambda: D.dp_train_on_batch(hd, [L, R], yd, mode="replicated")))
g = torch.zeros(295622, device="cuda")
print("all_reduce 1.18 MB %.4f ms" % t(lambda: dist.all_reduce(g)))
print("m[:2].cpu()        %.4f ms" % t(lambda: g[:2].cpu()))
dist.destroy_process_group()

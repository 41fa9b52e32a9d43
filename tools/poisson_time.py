import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd
from a_link_amd import noise as N
x = torch.randint(0, 256, (1024, 112, 112, 3), device="cuda").float()
xf = (x * 0.37 + 1.25)
for tag, t in (("integer pixels", x), ("fractional pixels", xf)):
    p = N.Poisson(seed=1)
    p.addNoise(t, None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): p.addNoise(t, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("poisson %s: %.2f ms / 1024 images, vals=%s" % (tag, dt * 1e3, p.last_vals[:2].tolist()))

# algorithmic bytes: 4 B read + 4 B written per element
x0 = torch.randint(0, 256, (1024, 112, 112, 3), device="cuda").float()
p = N.Poisson(seed=1)
for _ in range(2): p.addNoise(x0, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): p.addNoise(x0, None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("poisson integer pixels (events, incl. allocation of out/scratch): %.3f ms / 1024 images = %.2f TB/s algorithmic (8 B per element)" % (ms, x0.numel() * 8 / ms / 1e9))

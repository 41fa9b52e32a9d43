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

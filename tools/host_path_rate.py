#!/usr/bin/env python3
"""Embedding rate when the caller hands over HOST arrays (the reference's calling convention,
code/siamese.py:232-234): numpy float32 / uint8 pixels in, numpy embeddings out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import siamese

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4672
dt_ = sys.argv[2] if len(sys.argv) > 2 else None          # None: the API default (f16x2)
fm = siamese.ArcFace((112, 112), "synthetic:r100", dtype=dt_)
print("backbone dtype:", fm.model.model.dtype)
rng = np.random.RandomState(0)
x8 = rng.randint(0, 256, (n, 112, 112, 3)).astype(np.uint8)
xf = x8.astype(np.float32)
xd = torch.from_numpy(xf).cuda()
for name, arr in (("device f32 tensor", xd), ("host float32 array", xf), ("host uint8 array", x8)):
    fm.process(arr[:584]); fm.process(arr)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        fm.process(arr)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("%-20s %d images: %.1f ms -> %.0f embeddings/s" % (name, n, dt * 1e3, n / dt))

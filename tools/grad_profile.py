#!/usr/bin/env python3
"""Run embed_with_cache + input_gradient of the r100 teacher a few times (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a_link_amd  # noqa
from a_link_amd import siamese
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
conv = siamese.ArcFace((112, 112), "synthetic:r100", enable_grad=True, max_batch=n)
bb = conv.model.model
x = torch.randint(0, 256, (n, 112, 112, 3)).float().cuda()
d = torch.randn(n, 512, device="cuda")
for _ in range(3):
    bb.embed_with_cache(x)
    bb.input_gradient(d)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(5):
    bb.embed_with_cache(x)
torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 5
t = time.perf_counter()
for _ in range(5):
    bb.input_gradient(d)
torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 5
print("batch %d: forward+cache %.3f ms, backward %.3f ms" % (n, tf * 1e3, tb * 1e3))

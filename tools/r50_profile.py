#!/usr/bin/env python3
"""Per-op table of one VGGFace2 ResNet-50 forward (HIP events around every launch) + throughput."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import a_link_amd  # noqa: F401,E402
from a_link_amd.resnet50 import VGGResNet50  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt_ = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = VGGResNet50(max_batch=n, dtype=dt_)
print("dtype", dt_)
x = torch.randint(0, 256, (n, 224, 224, 3), device="cuda").float()
for _ in range(3):
    m.embed_device(x)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    m.embed_device(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
prof = m.profile(x)
tot = sum(ms for _, ms, _ in prof)
fl = sum(f for _, _, f in prof)
print("batch %d: %.3f ms/forward -> %.0f img/s ; %.1f TFLOP/s end to end ; sum of ops %.3f ms" % (n, dt * 1e3, n / dt, fl / dt / 1e12, tot))
print("%-26s %8s %8s" % ("op", "ms", "TF/s"))
for name, ms, f in prof:
    print("%-26s %8.3f %8.1f" % (name, ms, f / ms / 1e9 if ms > 0 else 0))

#!/usr/bin/env python3
"""Small-batch latency of the drop-in calls the reference makes per image / per pair:
FaceModel.get_feature (one CHW image), ArcFace.process (n images), SiameseNetwork.predict (n pairs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import a_link_amd  # noqa
from a_link_amd import siamese

_dt = [a.split("=")[1] for a in sys.argv if a.startswith("--dtype=")]
fm = siamese.ArcFace((112, 112), "synthetic:r100" + (":1:normalized" if "--normalized" in sys.argv else ""),
                     small_batch_split="--split" in sys.argv, dtype=_dt[0] if _dt else None)
print("backbone dtype:", fm.model.model.dtype)
net = siamese.SiameseNetwork((512,), "m", 0.1, seed=0)
rng = np.random.RandomState(0)


def med(fn, reps=30):
    fn(); fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


img = rng.randint(0, 256, (112, 112, 3)).astype(np.float32)
chw = fm.model.get_input(img)
print("FaceModel.get_feature (1 image, host in/out): %.3f ms" % med(lambda: fm.model.get_feature(chw)))
for n in (1, 4, 8, 16, 32, 64):
    x = rng.randint(0, 256, (n, 112, 112, 3)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    print("ArcFace.process n=%d: host arrays %.3f ms, device tensors %.3f ms" % (
        n, med(lambda: fm.process(x)), med(lambda: fm.process(xd))))
for n in (1, 16, 1024):
    L, R = rng.randn(n, 512).astype(np.float32), rng.randn(n, 512).astype(np.float32)
    print("SiameseNetwork.predict n=%d (host in/out): %.3f ms" % (n, med(lambda: net.predict([L, R]))))

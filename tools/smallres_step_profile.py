"""SmallRes 32 x 32 (reference code/siamese.py:134-184) train_on_batch, 16 pairs: N steps for `rocprofv3 --kernel-trace`, and the
wall clock per step (median, synchronised like the bench's smallres32_train_step_ms).

    python tools/smallres_step_profile.py [steps] [res] [feat]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import a_link_amd  # noqa: F401
from a_link_amd.smallres import SmallResNet


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    feat = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    srn = SmallResNet((res, res, 3), feat, lr=0.1, seed=1)
    rs = np.random.RandomState(0)
    L = torch.from_numpy(((rs.randint(0, 256, (16, res, res, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
    R = torch.from_numpy(((rs.randint(0, 256, (16, res, res, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
    y = torch.from_numpy(np.eye(2, dtype=np.float32)[rs.randint(0, 2, 16)]).cuda()
    np.random.seed(0)
    side = len(sys.argv) > 4 and sys.argv[4] == "side"          # A/B: the step on a stream of its own instead of torch's default (the NULL stream)
    ctx = torch.cuda.stream(torch.cuda.Stream()) if side else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(10):
            srn.train_on_batch([L, R], y)
        ts = []
        for _ in range(steps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            srn.train_on_batch([L, R], y)
            ts.append(time.perf_counter() - t)
    P = torch.from_numpy(((rs.randint(0, 256, (256, res, res, 3)) - 128.0) / 128.0).astype(np.float32)).cuda()
    srn.predict([P, P])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        srn.predict([P, P])
    torch.cuda.synchronize()
    tp = (time.perf_counter() - t) / 20
    print(json.dumps({"res": res, "feat": feat, "pairs": 16, "steps": steps, "train_step_ms_median": 1e3 * float(np.median(ts)),
                      "train_step_ms_mean": 1e3 * float(np.mean(ts)), "predict_256_pairs_ms": 1e3 * tp}))


if __name__ == "__main__":
    main()

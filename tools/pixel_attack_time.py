"""Few-pixel attack (code/attack.py:91-103 defaults: 40 pixels, population 200, 50 generations) on IR-100 at 112 x 112:
seconds per pair and backbone forwards/s, one pair after another against K pairs in lock-step.

    python tools/pixel_attack_time.py [pairs] [search] [lockstep ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import a_link_amd  # noqa: F401
from a_link_amd import attack as A, noise, siamese


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    search = sys.argv[2] if len(sys.argv) > 2 else "screen"
    widths = [int(v) for v in sys.argv[3:]] or [0, 32]
    conv = siamese.ArcFace((112, 112), "synthetic:r100")
    student = siamese.SiameseNetwork((512,), "/tmp/alink_student", 0.1, seed=1)
    wrapped = noise.PredictionWrappedModel(student, conv)
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, (224, 112, 3)).astype(np.float32) for _ in range(pairs)]
    targets = [[0, 1]] * pairs
    seeds = list(range(pairs))
    out = {"search": search, "pairs": pairs, "screening_dtype": getattr(A._device_parts(wrapped, search)[0], "dtype", None)}
    for K in widths:
        att = A.PixelAttacker(wrapped, search=search, lockstep=K)
        n = pairs if K else min(pairs, 4)
        if K == 0:
            att.attack_success = lambda *a, **k: None        # every generation runs, as when the attack does not succeed
        att.attack_all(imgs[:min(n, 2)], targets[:2], (224, 112), seeds=seeds[:2], maxiter=2, early_stop=False)   # warm-up
        torch.cuda.synchronize()
        t = time.perf_counter()
        att.attack_all(imgs[:n], targets[:n], (224, 112), seeds=seeds[:n], early_stop=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        res = att.last_results if K else None
        evals = sum(int(r.nfev) for r in res) if res else n * 51 * 200
        out["lockstep_%d" % K] = {"pairs": n, "s_per_pair": dt / n, "backbone_forwards_per_s": 2 * evals / dt}
        print(json.dumps({("lockstep_%d" % K): out["lockstep_%d" % K]}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

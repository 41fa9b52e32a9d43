#!/usr/bin/env python3
"""cProfile of the runnable driver (a-link_amd/ALINK_arc.py = reference code/ALINK_arc.py's __main__) on a synthetic DFW-style tree:
where the HOST spends its time around the kernels (loaders, pre-training, the framework loop)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import a_link_amd  # noqa: F401
from a_link_amd import ALINK_arc


def make_dfw(root, n_persons, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    d = os.path.join(root, "Training_data")
    for p in range(n_persons):
        pd = os.path.join(d, "person%03d" % p)
        os.makedirs(pd)
        for name, size in (("%03d.png" % p, (130, 120)), ("%03d_a.png" % p, (112, 112)), ("%03d_h_001.png" % p, (90, 100)),
                           ("%03d_h_002.png" % p, (150, 140)), ("%03d_I_001.png" % p, (112, 112))):
            Image.fromarray(rng.randint(0, 256, size + (3,)).astype(np.uint8)).save(os.path.join(pd, name))
    return root


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    tmp = tempfile.mkdtemp()
    root = make_dfw(tmp, n)
    models = os.path.join(tmp, "models")
    os.makedirs(models)
    common = ["--dataDirPrefix", root, "--arcface_model", "synthetic:r50:1:normalized", "--quiet",
              "--out_model", os.path.join(models, "postALINK"), "--ensemble_basepath", os.path.join(models, "ensemble"),
              "--disguised_basemodel", os.path.join(models, "disguisedModel"), "--pretrain_steps", "16000",
              "--dig_epochs", "1", "--undig_epochs", "1", "--noise", "gaussian,speckle,saltpepper,poisson"]
    for tag, extra in (("phase 1 (pre-train M2)", ["--train_disguised_model"]), ("phase 2 (ensemble + framework loop)", [])):
        np.random.seed(0)
        pr = cProfile.Profile()
        t = time.perf_counter()
        pr.enable()
        ALINK_arc.main(common + extra)
        pr.disable()
        print("==== %s: %.2f s, %d persons" % (tag, time.perf_counter() - t, n))
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()

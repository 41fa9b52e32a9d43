#!/usr/bin/env python3
"""Throughput of the kernels around the embedding path, one JSON object:
  * N x N verification score matrix (alink_pair_scores_matrix): pairs/s, f32-MFMA TFLOP/s
  * DFW protocol counts (alink_roc_counts): pairs/s and HBM GB/s (5 B per upper-triangle pair)
  * perturbation kernels (noise.hip): images/s and HBM GB/s (4 B read + 4 B written per element)
  * one generation of the few-pixel attack objective: 200 candidates -> 400 r100 embeddings + 200 pair scores
Run on the GPU box:  python tools/bench_aux.py [--arch r100]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=5, warm=1):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="r100")
    ap.add_argument("--n", type=int, default=7771)
    args = ap.parse_args()
    import numpy as np
    import torch
    import a_link_amd  # noqa: F401
    from a_link_amd import attack as A, evaluation as E, noise as N, siamese
    out = {}
    n = args.n
    for d in (512, 2048):
        net = siamese.SiameseNetwork((d,), "m", 0.1, seed=0)
        f = torch.randn(n, d, device="cuda")
        f = f / f.norm(dim=1, keepdim=True)
        S = torch.empty((n, n), device="cuda")
        dt = timeit(lambda: E.score_matrix(net, f, out=S), reps=2)
        macs = d * 512 + 512 * 64 + 64 * 2
        out["score_matrix_d%d" % d] = {"n": n, "s": dt, "pairs_per_s": n * n / dt, "tflops_f32": 2 * macs * n * n / dt / 1e12}
    M = torch.randint(0, 5, (n, n), device="cuda", dtype=torch.uint8)
    thr = np.linspace(0, 1, 1000)
    dt = timeit(lambda: E.roc_counts(S, M, thr, 3), reps=3)
    pairs = n * (n - 1) // 2
    out["roc_counts"] = {"n": n, "thresholds": 1000, "s": dt, "pairs_per_s": pairs / dt, "hbm_GBps": 5 * pairs / dt / 1e9}
    x = torch.randint(0, 256, (1024, 112, 112, 3), device="cuda").float()
    nbytes = 8 * x.numel()
    for name, obj in (("gaussian", N.Gaussian(seed=1)), ("speckle", N.Speckle(seed=1)), ("saltpepper", N.SaltPepper(seed=1)),
                      ("poisson", N.Poisson(seed=1))):
        dt = timeit(lambda: obj.addNoise(x, None), reps=5)
        out["noise_" + name] = {"images": 1024, "s": dt, "images_per_s": 1024 / dt, "hbm_GBps": nbytes / dt / 1e9}
    x224 = torch.randint(0, 256, (256, 224, 224, 3), device="cuda").float()
    p = N.Perlin(seed=1)
    dt = timeit(lambda: p.addNoise(x224, None), reps=5)
    out["noise_perlin_224"] = {"images": 256, "s": dt, "images_per_s": 256 / dt, "hbm_GBps": 8 * x224.numel() / dt / 1e9}
    dt = timeit(lambda: N.resize_images(x224, (112, 112)), reps=5)
    out["resize_224_to_112"] = {"images": 256, "s": dt, "images_per_s": 256 / dt}
    # one attack generation on the real network
    fm = siamese.ArcFace((112, 112), "synthetic:%s" % args.arch)
    pm = siamese.SiameseNetwork((512,), "m2", 0.1, seed=4)
    wrapped = N.PredictionWrappedModel(pm, fm)
    img = np.random.RandomState(0).randint(0, 256, (224, 112, 3)).astype(np.float32)
    sc = A._DevicePairScorer(wrapped, img)
    xs = np.random.RandomState(1).rand(200, 200) * np.tile([224, 112, 256, 256, 256], 40)
    dt = timeit(lambda: sc.predict(xs), reps=10, warm=2)
    out["attack_generation"] = {"arch": args.arch, "candidates": 200, "s": dt, "embeddings_per_s": 400 / dt}
    att = A.PixelAttacker(wrapped, seed=np.random.RandomState(3))
    t = time.perf_counter()
    att.attack(img, 1, 0, pixel_count=40, dimensions=(224, 112), maxiter=50, popsize=250)
    dt = time.perf_counter() - t
    r = att.last_result
    out["attack_full"] = {"arch": args.arch, "s": dt, "generations": int(r.nit), "evaluations": int(r.nfev),
                          "backbone_forwards": int(2 * r.nfev), "s_per_generation": dt / max(1, int(r.nit))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

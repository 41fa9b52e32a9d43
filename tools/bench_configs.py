#!/usr/bin/env python3
"""One-GPU measurements of BASELINE.json configs[2..4] (the per-GPU share of the 8-GPU shapes), one
JSON object on stdout.  The headline config is bench.py's; this tool adds the surrounding workloads
(SURVEY.md §8d synthetic inputs), each checked for finiteness, none of them a substitute for bench.py.

  C3  committee of 3 IR-ResNet-50 backbones + 3 pair heads scoring a pool shard of 12,500 images
      (100k / 8 GPUs) against a 16-image gallery: embed x3, 200k pairs x3 heads with the committee
      mean fused, entropy, exact top-1024.
  C4  one A-LINK iteration (code/ALINK_arc.py:142-254) with an IR-ResNet-100 teacher: 16 persons,
      2 plain + 3 disguised images each (P = 3840 pairs, 80 unique images), noises
      gaussian/saltpepper/poisson/speckle drawn per pair occurrence, selection, fine-tune.
  C5  the A2-LINK adversarial noise: few-pixel differential-evolution attack (40 pixels, popsize 200,
      <= 50 generations) on r100 pairs; reported per pair (an iteration has thousands of pairs).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sync_time(fn, reps=1):
    import torch
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, r


DTYPE = None        # --dtype: None = the API default (face_model.default_dtype: "f16x2", the selection mode)


def config3(pool_n, out):
    """The product path bench.py times under torchrun (distributed.committee_pool_topk), at the full per-GPU shard."""
    import torch
    from a_link_amd import distributed as D, siamese
    members = [siamese.ArcFace((112, 112), "synthetic:r50:%d" % s, dtype=DTYPE) for s in (1, 2, 3)]
    heads = [siamese.SiameseNetwork((512,), "h%d" % i, 0.1, seed=i) for i in range(3)]
    g = torch.Generator().manual_seed(0)
    pool = torch.randint(0, 256, (pool_n, 112, 112, 3), generator=g, dtype=torch.uint8).cuda()
    gallery = torch.randint(0, 256, (16, 112, 112, 3), generator=g, dtype=torch.uint8).cuda()
    bbs = [m.model.model for m in members]
    hds = [h.siamese_net for h in heads]

    def run():
        return D.committee_pool_topk(bbs, hds, pool, gallery, 1024, shard_offset=0)
    run()
    dt, (vals, idx) = sync_time(run, 2)
    assert bool(torch.isfinite(vals).all()) and idx.numel() == 1024
    out["config3_committee_pool_shard"] = {
        "pool_images_per_gpu": pool_n, "members": 3, "arch": "r50", "gallery": 16, "pairs": pool_n * 16, "s": dt,
        "pool_images_per_s": pool_n / dt, "backbone_forwards_per_s": 3 * (pool_n + 16) / dt}
    del members, heads


def _people(n, k, seed):
    import numpy as np
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 256, (k, 112, 112, 3)).astype(np.float32) for _ in range(n)]


def config4(out, noises, student_dtype="f32"):
    import numpy as np
    import torch
    from a_link_amd import alink_loop as AL, committee, noise, pairs, siamese
    conv = siamese.ArcFace((112, 112), "synthetic:r100", dtype=DTYPE)
    student = siamese.SiameseNetwork((512,), "/tmp/alink_student", 0.1, seed=1, compute_dtype=student_dtype)
    ens = [siamese.SiameseNetwork((512,), "e1", 0.1, seed=2)]
    np.random.seed(0)                                    # the noise objects draw their Philox seeds at construction
    nz = [noise.get_relevant_noise(n)(model=student, sess=None, feature_model=conv) for n in noises]
    bag = committee.Bagging(ens, nz)
    X_plain, X_dig = _people(16, 2, 1), _people(16, 3, 2)
    feats = [conv.process(p) for p in X_plain]
    gen = pairs.getGenerator(pairs.getNormalGenerator(feats, 16), pairs.getNormalGenerator(feats, 16),
                             pairs.getImposterGenerator(feats, feats, 16), 16)
    flags = AL.Flags(out_model="", eps=0.0005)
    np.random.seed(0)
    AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, (112, 112), col=0, verbose=0)   # warm-up
    torch.cuda.synchronize()
    t = time.perf_counter()
    st = AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, (112, 112), col=0, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    P = st.un_size
    out["config4_alink_iteration_r100" + ("" if student_dtype == "f32" else "_student_" + student_dtype)] = {
        "student_compute_dtype": student_dtype, "persons": 16, "unique_images": 80, "pairs": P, "noises": list(noises), "s": dt,
        "backbone_forwards": 80 + 2 * P * len(noises), "backbone_forwards_per_s": (80 + 2 * P * len(noises)) / dt,
        "oracle_queries": st.active_count, "finetunes": st.finetunes,
        "reference_shape_forwards": 2 * P * (1 + len(noises))}
    return conv, student


def config5(out, conv, student, n_pairs, search="exact"):
    """the few-pixel attack at the reference's defaults, every search run to maxiter: K pairs in lock-step (round 6), and two pairs one
    after another (the reference's shape) beside it"""
    import numpy as np
    from a_link_amd import attack as A, noise
    wrapped = noise.PredictionWrappedModel(student, conv)
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, (224, 112, 3)).astype(np.float32) for _ in range(n_pairs)]
    targets = [[0, 1]] * n_pairs
    seeds = list(range(n_pairs))
    att = A.PixelAttacker(wrapped, search=search)
    att.attack_all(imgs[:2], targets[:2], (224, 112), seeds=seeds[:2], maxiter=2, early_stop=False)      # warm-up
    t = time.perf_counter()
    att.attack_all(imgs, targets, (224, 112), seeds=seeds, early_stop=False)
    dt = time.perf_counter() - t
    evals = sum(int(r.nfev) for r in att.last_results)
    key = "config5_pixel_attack_r100" + ("" if search == "exact" else "_search_" + search)
    out[key] = {"pairs": n_pairs, "lockstep": att.lockstep, "s_per_pair": dt / n_pairs,
                "generations_per_pair": float(np.mean([r.nit for r in att.last_results])),
                "backbone_forwards_per_pair": 2 * evals / n_pairs, "backbone_forwards_per_s": 2 * evals / dt}
    seq = A.PixelAttacker(wrapped, search=search, lockstep=0)
    seq.attack_success = lambda *a, **k: None            # every generation runs, as when the attack does not succeed
    t = time.perf_counter()
    seq.attack_all(imgs[:2], targets[:2], (224, 112), seeds=seeds[:2])
    out[key]["one_pair_after_another_s_per_pair"] = (time.perf_counter() - t) / 2


def config5_gradient(out, n_pairs):
    """FGSM / PGD (extension): signed-gradient steps on both images of every pair through the r100 teacher."""
    import numpy as np
    import torch
    from a_link_amd import noise, siamese
    conv = siamese.ArcFace((112, 112), "synthetic:r100", enable_grad=True, max_batch=256)
    student = siamese.SiameseNetwork((512,), "/tmp/alink_student", 0.1, seed=1)
    g = torch.Generator().manual_seed(0)
    L = torch.randint(0, 256, (n_pairs, 112, 112, 3), generator=g).float().cuda()
    R = torch.randint(0, 256, (n_pairs, 112, 112, 3), generator=g).float().cuda()
    target = np.random.RandomState(0).randint(0, 2, n_pairs)
    bb = conv.model.model
    nb = bb.max_batch
    x = L[:nb]
    bb.embed_with_cache(x)
    d = torch.randn(nb, 512, device="cuda")
    dt_f, _ = sync_time(lambda: bb.embed_with_cache(x), 5)
    dt_b, _ = sync_time(lambda: bb.input_gradient(d), 5)
    out["config5_input_gradient_r100"] = {"batch": nb, "forward_with_cache_ms": dt_f * 1e3, "backward_ms": dt_b * 1e3,
                                          "gradient_images_per_s": nb / (dt_f + dt_b)}
    for name, att in (("fgsm", noise.FGSM(model=student, feature_model=conv, eps=4.0)),
                      ("pgd5", noise.PGD(model=student, feature_model=conv, eps=4.0, alpha=1.0, steps=5, seed=1))):
        att.addPairNoise([L[:nb], R[:nb]], target[:nb])
        dt, (al, ar) = sync_time(lambda: att.addPairNoise([L, R], target), 1)
        assert bool(torch.isfinite(al).all()) and float((al - L).abs().max()) <= 4.0 + 1e-3
        out["config5_%s_r100" % name] = {"pairs": n_pairs, "s": dt, "pairs_per_s": n_pairs / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=12500)
    ap.add_argument("--attack-pairs", type=int, default=3)
    ap.add_argument("--noises", default="gaussian,saltpepper,poisson,speckle")
    ap.add_argument("--skip", default="")
    ap.add_argument("--dtype", default=None, help="backbone storage of the feature models (default: the API default, f16x2 = the "
                    "selection mode; bf16 = the screening mode)")
    a = ap.parse_args()
    import a_link_amd  # noqa: F401
    from a_link_amd import face_model
    global DTYPE
    DTYPE = a.dtype
    out = {"backbone_dtype": a.dtype or face_model.default_dtype()}
    if "3" not in a.skip:
        config3(a.pool, out)
    conv = student = None
    if "4" not in a.skip:
        conv, student = config4(out, a.noises.split(","))
        del conv, student
        # configs[4]: the same iteration with the student fine-tuned in the head's bf16 compute mode
        conv, student = config4(out, a.noises.split(","), student_dtype="bf16")
    if "5" not in a.skip:
        if conv is None:
            from a_link_amd import siamese
            conv = siamese.ArcFace((112, 112), "synthetic:r100", dtype=DTYPE)
            student = siamese.SiameseNetwork((512,), "/tmp/alink_student", 0.1, seed=1)
        config5(out, conv, student, a.attack_pairs)
        if getattr(conv, "screen", None) is not None:      # the same searches with their candidates in the screening form
            config5(out, conv, student, a.attack_pairs, search="screen")
    if "g" not in a.skip:
        del conv, student
        config5_gradient(out, 1024)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

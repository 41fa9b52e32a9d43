#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[2] / SURVEY.md §8d C3 at its real depth: a committee of THREE IR-50
backbones + three pair heads scoring a 2,048-image pool subsample against a fixed 16-image gallery at 112x112 —
computed by the CPU ORACLE (oracle/ir_resnet.py, oracle/siamese_head.py, oracle/al_logic.py; ~6,200 float32
IR-50 forwards, a few minutes on this container's 8 cores; a committed fixture because the GPU box's host leg
should not spend them, and so that the expected numbers do not come from the machine under test).

    python tests/golden/make_golden_config3.py          -> tests/golden/config3_r50.npz

Everything the test needs to rebuild the SAME inputs is integer / seeded NumPy (no file travels but the fixture):
pixels    64 synthetic identities: a blocky random base face (8x8-pixel blocks, uint8) + per-image integer noise,
          32 pool images each (2,048), and one further image of each of the first 16 as the gallery — so that
          (pool, gallery) pairs range from "same person" to "unrelated", like an unlabeled pool against enrolled
          faces; all integer arithmetic, bit-reproducible on any machine.
backbones W.synthetic_ir_params(R50_UNITS, seed=s), s = 1,2,3 (conv He-normal, BN gamma U(.5,1.5) ...),
          with every BatchNorm's moving mean / variance CALIBRATED to its input (oracle/calibrate.py) the way a
          trained checkpoint's are, then rounded to float16 and stored here: the stored values ARE the weights, so
          the GPU box loads bit-identical tensors although calibration itself is float arithmetic.
heads     oracle.siamese_head.init_weights(512, seed=10+m) with the last layer rescaled (W3 *= gain[m],
          b3 = [0, bias[m]]; stored) so that member probabilities spread over (0,1) instead of sitting at
          0.5 +- 0.02 as a fresh glorot head's do.
Stored: the committee mean `ens` (32768,2) float32 = Bagging.predict (reference code/committee.py:13-20) over pairs
(pool i, gallery j) in i-major order, its top-1024 by entropy, the members' gallery embeddings and first 8 pool
embeddings (run-time spot check of the fixture against the oracle code), BN statistics, gains / biases.
The reference itself holds no fixture for this path: the CNN oracle stays "parity unpinned".
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import a_link_amd  # noqa: E402,F401
from a_link_amd import weights as W  # noqa: E402
from oracle import al_logic as OA  # noqa: E402
from oracle import ir_resnet  # noqa: E402
from oracle import siamese_head as O  # noqa: E402

N_PERSONS, PER_PERSON, N_GAL = 64, 32, 16
N_POOL = N_PERSONS * PER_PERSON


def inputs():
    """-> pool (2048,112,112,3) uint8, gallery (16,112,112,3) uint8, li, ri (pool-major pair lists)."""
    rng = np.random.default_rng(0)                                            # pool seed 0
    coarse = rng.integers(40, 216, (N_PERSONS, 14, 14, 3), dtype=np.int16)
    bases = np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)                # 112 x 112 blocky "faces"
    noise = rng.integers(-40, 41, (N_PERSONS, PER_PERSON + 1, 112, 112, 3), dtype=np.int16)
    shade = rng.integers(-20, 21, (N_PERSONS, PER_PERSON + 1, 1, 1, 1), dtype=np.int16)
    imgs = np.clip(bases[:, None] + noise + shade, 0, 255).astype(np.uint8)
    pool = imgs[:, :PER_PERSON].reshape(N_POOL, 112, 112, 3)
    gallery = imgs[:N_GAL, PER_PERSON]
    li = np.repeat(np.arange(N_POOL, dtype=np.int32), N_GAL)
    ri = np.tile(np.arange(N_GAL, dtype=np.int32), N_POOL)
    return np.ascontiguousarray(pool), np.ascontiguousarray(gallery), li, ri


def bn_stat_names(params):
    return sorted(k for k in params if k.endswith("_moving_mean") or k.endswith("_moving_var"))


def member_params(seed, stats=None):
    """The member's backbone tensors; `stats` (flat float16 vector in bn_stat_names order) = the stored statistics."""
    params = W.synthetic_ir_params(W.R50_UNITS, seed=seed)
    if stats is not None:
        o = 0
        for k in bn_stat_names(params):
            n = params[k].size
            params[k] = stats[o:o + n].astype(np.float32)
            o += n
        assert o == len(stats)
    return params


def spread_head(seed, d_pairs):
    """glorot head whose last layer is rescaled so that the logit difference runs from about -2 to +2 (p from 0.12 to
    0.88) between the 1st and the 99th percentile of `d_pairs` = (L, R) embedding rows, centred between them —
    instead of the +-0.08 (p = 0.5 +- 0.02) of a fresh glorot head."""
    ws = O.init_weights(512, seed=seed)
    _, (d, z1, a1, z2, a2) = O.forward(ws, d_pairs[0], d_pairs[1], cache=True)
    t = (a2 @ (ws[4][:, 1] - ws[4][:, 0])).astype(np.float64)
    lo, hi = np.percentile(t, [1, 99])
    gain = np.float32(4.0 / max(hi - lo, 1e-12))
    ws[4] = (ws[4] * gain).astype(np.float32)
    bias = np.float32(-gain * 0.5 * (lo + hi))
    ws[5] = np.array([0, bias], np.float32)
    return ws, gain, bias


def head_weights(seed, gain, bias):
    ws = O.init_weights(512, seed=seed)
    ws[4] = (ws[4] * np.float32(gain)).astype(np.float32)
    ws[5] = np.array([0, bias], np.float32)
    return ws


def main():
    from oracle import calibrate
    pool, gallery, li, ri = inputs()
    out = {}
    members = []
    t0 = time.time()
    for m, seed in enumerate((1, 2, 3)):
        params = member_params(seed)
        calibrate.calibrate_(params, pool[m::128].astype(np.float32))          # 16 pool images, different per member
        stats = np.concatenate([params[k].ravel() for k in bn_stat_names(params)]).astype(np.float16)
        var_pos = np.concatenate([np.full(params[k].size, k.endswith("_var")) for k in bn_stat_names(params)])
        assert (stats[var_pos] > 0).all()
        params = member_params(seed, stats)                                   # what the test will load
        Eg = ir_resnet.embed(params, gallery.astype(np.float32))
        Ep = np.zeros((N_POOL, 512), np.float32)
        for i in range(0, N_POOL, 64):
            Ep[i:i + 64] = ir_resnet.embed(params, pool[i:i + 64].astype(np.float32))
            print("member %d: %d / %d  (%.0f s)" % (m, i + 64, N_POOL, time.time() - t0), flush=True)
        ws, gain, bias = spread_head(10 + m, (Ep[li], Eg[ri]))
        members.append(O.forward(ws, Ep[li], Eg[ri]))
        out["bn_stats_%d" % m] = stats
        out["gallery_emb_%d" % m] = Eg
        out["pool_emb_head_%d" % m] = Ep[:8]
        out["gain_%d" % m], out["bias_%d" % m] = gain, bias
        out["member_probs_head_%d" % m] = members[-1][:8 * N_GAL]
        c = Ep @ Eg.T
        same = c[np.arange(N_GAL * PER_PERSON) // PER_PERSON * 0 + np.arange(N_GAL * PER_PERSON), np.arange(N_GAL * PER_PERSON) // PER_PERSON]
        print("member %d: cos(pool, gallery) same person %.2f +- %.2f, others %.2f +- %.2f, gain %.1f"
              % (m, same.mean(), same.std(), c[N_GAL * PER_PERSON:].mean(), c[N_GAL * PER_PERSON:].std(), gain))
    ens = OA.bagging_predict(members).astype(np.float32)
    out["ens"] = ens
    ent = OA.proba_entropy(ens)
    order = np.lexsort((np.arange(len(ent)), -ent))
    out["top1024"] = order[:1024].astype(np.int32)
    print("p[:,0] percentiles 0/10/25/50/75/90/100", np.percentile(ens[:, 0], [0, 10, 25, 50, 75, 90, 100]).round(3))
    print("|p - 0.5| at the cut", np.sort(np.abs(ens[:, 0] - 0.5))[1023])
    np.savez_compressed(os.path.join(HERE, "config3_r50.npz"), **out)
    print("fixture bytes", os.path.getsize(os.path.join(HERE, "config3_r50.npz")))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[2] / SURVEY.md §8d C3 at its real depth: a committee of THREE IR-50
backbones (SURVEY synthetic weights, seeds 1,2,3) + three pair heads scoring a 2,048-image pool subsample
against a fixed 16-image gallery at 112x112 — computed by the CPU ORACLE (oracle/ir_resnet.py,
oracle/siamese_head.py, oracle/al_logic.py; ~6,200 float32 IR-50 forwards, about 40 minutes on this
container's 8 cores, which is why it is a committed fixture and not run-time work on the GPU box).

    python tests/golden/make_golden_config3.py          -> tests/golden/config3_r50.npz

Everything the test needs to rebuild the SAME inputs is seeded NumPy (no file travels but the fixture):
pixels  pool    = default_rng(0).integers(0, 256, (2048,112,112,3), uint8)          (pool seed 0)
        gallery = default_rng(100).integers(0, 256, (16,112,112,3), uint8)
weights backbones W.synthetic_ir_params(R50_UNITS, seed=s) for s in 1,2,3
        heads     oracle.siamese_head.init_weights(512, seed=10+m) with the last layer rescaled
                  (W3 *= gain[m], b3 = bias[m]; gain/bias stored here) so that the member's probabilities
                  spread over (0,1) instead of sitting at 0.5 +- 0.02 as a fresh glorot head's do.
Stored: the committee mean `ens` (32768,2) float32 = Bagging.predict (reference code/committee.py:13-20) over
pairs (pool i, gallery j) in i-major order, the members' gallery embeddings (for a run-time spot check of the
fixture against the oracle code on a few pool images), gains/biases.
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

N_POOL, N_GAL = 2048, 16


def inputs():
    pool = np.random.default_rng(0).integers(0, 256, (N_POOL, 112, 112, 3), dtype=np.uint8)
    gallery = np.random.default_rng(100).integers(0, 256, (N_GAL, 112, 112, 3), dtype=np.uint8)
    li = np.repeat(np.arange(N_POOL, dtype=np.int32), N_GAL)
    ri = np.tile(np.arange(N_GAL, dtype=np.int32), N_POOL)
    return pool, gallery, li, ri


def spread_head(seed, d_pairs):
    """glorot head whose last layer is rescaled so logit differences have median 0 and an inter-decile
    range of about +-2 (p from ~0.12 to ~0.88) on `d_pairs` = (L, R) embedding rows."""
    ws = O.init_weights(512, seed=seed)
    _, (d, z1, a1, z2, a2) = O.forward(ws, d_pairs[0], d_pairs[1], cache=True)
    t = (a2 @ (ws[4][:, 1] - ws[4][:, 0])).astype(np.float64)
    lo, med, hi = np.percentile(t, [10, 50, 90])
    gain = np.float32(4.0 / max(hi - lo, 1e-12))
    bias = np.float32(-gain * med)
    ws[4] = (ws[4] * gain).astype(np.float32)
    ws[5] = np.array([0, bias], np.float32)
    return ws, gain, bias


def main():
    pool, gallery, li, ri = inputs()
    out = {}
    members = []
    t0 = time.time()
    for m, seed in enumerate((1, 2, 3)):
        params = W.synthetic_ir_params(W.R50_UNITS, seed=seed)
        Eg = ir_resnet.embed(params, gallery.astype(np.float32))
        Ep = np.zeros((N_POOL, 512), np.float32)
        for i in range(0, N_POOL, 64):
            Ep[i:i + 64] = ir_resnet.embed(params, pool[i:i + 64].astype(np.float32))
            print("member %d: %d / %d  (%.0f s)" % (m, i + 64, N_POOL, time.time() - t0), flush=True)
        ws, gain, bias = spread_head(10 + m, (Ep[li], Eg[ri]))
        members.append(O.forward(ws, Ep[li], Eg[ri]))
        out["gallery_emb_%d" % m] = Eg
        out["pool_emb_head_%d" % m] = Ep[:8]
        out["gain_%d" % m], out["bias_%d" % m] = gain, bias
        out["member_probs_head_%d" % m] = members[-1][:8 * N_GAL]
    ens = OA.bagging_predict(members).astype(np.float32)
    out["ens"] = ens
    ent = OA.proba_entropy(ens)
    order = np.lexsort((np.arange(len(ent)), -ent))
    out["top1024"] = order[:1024].astype(np.int32)
    print("p[:,0] deciles", np.percentile(ens[:, 0], [0, 10, 25, 50, 75, 90, 100]))
    print("entropy cut", ent[order[1023]], "max", ent.max())
    np.savez_compressed(os.path.join(HERE, "config3_r50.npz"), **out)


if __name__ == "__main__":
    main()

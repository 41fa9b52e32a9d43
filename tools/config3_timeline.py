#!/usr/bin/env python3
"""Where one screen-then-settle config-3 pass (bench.py's leg: 3 x IR-50 committee, 12,500 pool images x 16 gallery images, entropy,
top-1024) spends its wall clock: cProfile of distributed.committee_pool_topk_settled, cumulative times of its stages."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import a_link_amd  # noqa: F401
import bench
from a_link_amd import distributed as D, weights as W
from a_link_amd.backbone import IRBackbone
from a_link_amd.head import DenseHead


def main():
    n_shard = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    cal, gal = bench._identity_pool(512, 999)
    shard, _ = bench._identity_pool(n_shard, 1000)
    exa, scr = [], []
    for s_ in (1, 2, 3):
        pr = W.synthetic_ir_params(W.R50_UNITS, seed=s_, normalized=True)
        e_ = IRBackbone(pr, dtype="f16x2", max_batch=292, streams=2)
        e_.calibrate(cal[:292])
        exa.append(e_)
        scr.append(IRBackbone(pr, dtype="f16", max_batch=292, streams=2))
    lic = torch.arange(512, dtype=torch.int32, device="cuda").repeat_interleave(16)
    ric = torch.arange(16, dtype=torch.int32, device="cuda").repeat(512)
    Ecal = [e_.embed_device(cal) for e_ in exa]
    Egal = [e_.embed_device(gal) for e_ in exa]
    yc = (lic.cpu().numpy() // 32 == ric.cpu().numpy())
    rs = np.random.RandomState(0)
    pick = np.concatenate([np.flatnonzero(yc), rs.choice(np.flatnonzero(~yc), 3 * int(yc.sum()), replace=False)])
    rs.shuffle(pick)
    yoh = np.stack([~yc[pick], yc[pick]], 1).astype(np.float32)
    heads = []
    for m_ in range(3):
        h_ = DenseHead(512, lr=1.0, seed=10 + m_)
        np.random.seed(100 + m_)
        h_.fit([Ecal[m_].cpu().numpy()[lic.cpu().numpy()[pick]], Egal[m_].cpu().numpy()[ric.cpu().numpy()[pick]]], yoh, batch_size=64, epochs=6, verbose=0)
        heads.append(h_)
    D.committee_pool_topk_settled(scr, exa, heads, shard[:584], gal, 64, 0)          # warm-up
    for name, fn in (("screening only", lambda: D.committee_pool_topk(scr, heads, shard, gal, 1024, 0)),
                     ("all exact", lambda: D.committee_pool_topk(exa, heads, shard, gal, 1024, 0))):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print("%-16s %.1f ms" % (name, 1e3 * (time.perf_counter() - t)))
    info = {}
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t = time.perf_counter()
    pr.enable()
    D.committee_pool_topk_settled(scr, exa, heads, shard, gal, 1024, 0, info=info)
    torch.cuda.synchronize()
    pr.disable()
    print("screen-then-settle %.1f ms  %s" % (1e3 * (time.perf_counter() - t), {k: info[k] for k in ("images_settled", "rounds", "delta", "audit")}))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()

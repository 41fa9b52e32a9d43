#!/usr/bin/env python3
"""Where one A-LINK iteration (bench.py's config-4 leg, screen-then-settle) spends its wall clock: every stage of
alink_loop.alink_iteration wrapped in a synchronising timer (so the sum exceeds the un-instrumented iteration slightly)."""
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import a_link_amd  # noqa: F401
import bench
from a_link_amd import alink_loop as AL, committee, noise as NZ, pairs as PR, settle, siamese

T = OrderedDict()


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t
        return r
    return w


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "screen_settle"
    names = ("gaussian", "saltpepper", "poisson", "speckle")
    ppl, _ = bench._identity_pool(16 * 5, 4242, per_person=5)
    ppl = ppl.float().cpu().numpy().reshape(16, 5, 112, 112, 3)
    X_plain, X_dig = [q[:2] for q in ppl], [q[2:] for q in ppl]
    conv = siamese.ArcFace((112, 112), "synthetic:r100:1:normalized", dtype="f16x2", screen_dtype="auto" if mode == "screen_settle" else None)
    conv.model.model.calibrate(torch.from_numpy(ppl.reshape(-1, 112, 112, 3)).cuda())
    student = siamese.SiameseNetwork((512,), "/tmp/alink_student", 0.1, seed=1)
    ens = [siamese.SiameseNetwork((512,), "e%d" % i, 0.1, seed=2 + i) for i in range(2)]
    nzs = [NZ.get_relevant_noise(n_)(model=student, sess=None, feature_model=conv) for n_ in names]
    feats = [conv.process(q) for q in X_plain]
    allf = torch.from_numpy(np.concatenate(feats)).cuda()
    li4 = torch.arange(len(allf), dtype=torch.int32, device="cuda").repeat_interleave(len(allf))
    ri4 = torch.arange(len(allf), dtype=torch.int32, device="cuda").repeat(len(allf))
    for h_ in [student] + ens:
        bench._spread_head(h_.siamese_net, allf, allf, li4, ri4)
    bag = committee.Bagging(ens, nzs)
    flags = AL.Flags(out_model="", screen_settle=mode == "screen_settle")
    for rep in range(2):
        if rep == 1:
            conv.process = timed("process (exact embed)", conv.process)
            if mode == "screen_settle":
                conv.process_screen = timed("process_screen (16-bit embed)", conv.process_screen)
            bag.attackModel = timed("attackModel (noise + resize kernels)", bag.attackModel)
            bag.predict = timed("bag.predict (clean pairs)", bag.predict)
            student.predict = timed("student.predict", student.predict)
            student.finetune = timed("student.finetune", student.finetune)
            o = settle.select_queries_settled
            settle.select_queries_settled = timed("select_queries_settled (incl. its settles)", o)
            AL.selection.select_queries = timed("select_queries", AL.selection.select_queries)
            AL._embed_pairs_unique = timed("_embed_pairs_unique (incl. clean embed)", AL._embed_pairs_unique)
        for i_, z in enumerate(nzs):
            z._seed, z._calls = 1000 + i_, 0
        np.random.seed(0)
        gen = PR.getGenerator(PR.getNormalGenerator(feats, 16), PR.getNormalGenerator(feats, 16), PR.getImposterGenerator(feats, feats, 16), 16)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = AL.run_alink_dfw(flags, conv, bag, nzs, student, X_plain, X_dig, gen, (112, 112), col=0, verbose=0)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t1
    print("mode %s: iteration %.3f s (instrumented), pairs %d, queries %d, finetunes %d" % (mode, tot, st.un_size, st.active_count, st.finetunes))
    for k, v in T.items():
        print("  %-48s %7.1f ms" % (k, 1e3 * v))
    if st.settle_info:
        print("  settle info:", st.settle_info[-1])


if __name__ == "__main__":
    main()

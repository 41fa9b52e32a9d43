"""One A2-LINK iteration WITH the reference's default noise list's black-box member (code/ALINK_arc.py:41: `adversarial` =
the few-pixel attack of code/attack.py:91-103 on every pair of the mini-batch) on an IR-100 teacher: wall clock of the whole
iteration and of the attack inside it.  The reference's mini-batch is 16 persons (3,840 pairs: 29 min of attack on one GPU at
0.456 s per pair); this tool runs `persons` of them (default 6: 540 pairs) so that one call fits a GPU lease, and says what
the full size extrapolates to.

    python tools/a2link_iteration.py [persons] [search]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import a_link_amd  # noqa: F401
from a_link_amd import alink_loop as AL, committee, noise, pairs, siamese


def main():
    persons = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    search = sys.argv[2] if len(sys.argv) > 2 else "screen"
    rng = np.random.RandomState(1)
    X_plain = [rng.randint(0, 256, (2, 112, 112, 3)).astype(np.float32) for _ in range(persons)]
    X_dig = [rng.randint(0, 256, (3, 112, 112, 3)).astype(np.float32) for _ in range(persons)]
    conv = siamese.ArcFace((112, 112), "synthetic:r100:1:normalized")
    conv.calibrate(np.concatenate(X_plain + X_dig))
    student = siamese.SiameseNetwork((512,), "/tmp/alink_student", 0.1, seed=1)
    ens = [siamese.SiameseNetwork((512,), "e1", 0.1, seed=2)]
    np.random.seed(0)
    cheap = [noise.get_relevant_noise(n)(model=student, sess=None, feature_model=conv) for n in ("gaussian", "saltpepper", "poisson", "speckle")]
    adv = noise.AdversarialNoise(student, None, conv, search=search)
    spent = {"attack_s": 0.0, "pairs": 0}
    inner = adv.addPairNoise

    def timed(image_pairs, target_labels, rows=None):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = inner(image_pairs, target_labels, rows=rows)
        torch.cuda.synchronize()
        spent["attack_s"] += time.perf_counter() - t
        spent["pairs"] += len(image_pairs[0])
        return r
    adv.addPairNoise = timed
    nz = cheap + [adv]
    bag = committee.Bagging(ens, nz)
    feats = [conv.process(p) for p in X_plain]
    gen = pairs.getGenerator(pairs.getNormalGenerator(feats, 16), pairs.getNormalGenerator(feats, 16),
                             pairs.getImposterGenerator(feats, feats, 16), 16)
    flags = AL.Flags(out_model="", eps=0.0005, alink_bs=persons)
    np.random.seed(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    st = AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, (112, 112), col=0, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    P = st.un_size
    res = getattr(adv.attacker, "last_results", None) or []
    fwd = sum(2 * int(r.nfev) for r in res)
    out = {"persons": persons, "pairs": P, "noises": "gaussian,saltpepper,poisson,speckle,adversarial", "search": search,
           "iteration_s": dt, "attack_s": spent["attack_s"], "everything_else_s": dt - spent["attack_s"],
           "attack_s_per_pair": spent["attack_s"] / max(spent["pairs"], 1),
           "attack_backbone_forwards": fwd, "attack_backbone_forwards_per_s": fwd / max(spent["attack_s"], 1e-9),
           "mean_generations_per_pair": float(np.mean([r.nit for r in res])) if res else None,
           "oracle_queries": st.active_count, "finetunes": st.finetunes,
           "reference_batch_extrapolation": {"pairs": 3840, "attack_min": 3840 * spent["attack_s"] / max(spent["pairs"], 1) / 60.0}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""customTrainModel (reference code/siamese.py:81-112) at the reference's shape: 200 persons, head-512, batch 16 — ms per step
of the three forms (index batches / host gathers with deferred metrics / synchronised step by step) and where the host time of
the indexed form goes.

    python tools/custom_train_time.py [steps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import a_link_amd  # noqa: F401
from a_link_amd import pairs, siamese


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    only = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. "indexed": that form alone (for a kernel trace)
    rng = np.random.RandomState(0)
    feats = [rng.randn(rng.randint(3, 6), 512).astype(np.float32) for _ in range(200)]

    def make():
        return pairs.getGenerator(pairs.getNormalGenerator(feats, 16), pairs.getNormalGenerator(feats, 16),
                                  pairs.getImposterGenerator(feats, feats, 16), 16)
    out = {"persons": 200, "D": 512, "batch_size": 16, "steps": steps}
    logs = {}
    for name, indexed, deferred, n in (("indexed", True, True, steps), ("host_gather_deferred", False, True, steps),
                                       ("step_by_step", False, False, max(200, steps // 4))):
        if only and name != only:
            continue
        net = siamese.SiameseNetwork((512,), "/tmp/ctm", 0.1, seed=1)
        net._index_steps, net._defer_metrics = indexed, deferred
        gen = make()
        np.random.seed(0)
        net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 300, verbose=0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        logs[name] = net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * n, verbose=0)
        torch.cuda.synchronize()
        out[name + "_ms_per_step"] = 1e3 * (time.perf_counter() - t) / n
    if only:
        print(json.dumps(out))
        return
    out["logs_indexed_equal_host_gather"] = logs["indexed"] == logs["host_gather_deferred"]
    # the host side of the indexed form alone: the whole planning loop with the launch call stubbed out ...
    class _NoLaunch(object):
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            if name == "alink_head_custom_train_steps":
                return lambda *a: 0
            return getattr(self._lib, name)
    net = siamese.SiameseNetwork((512,), "/tmp/ctm", 0.1, seed=1)
    net.siamese_net.lib = _NoLaunch(net.siamese_net.lib)
    gen = make()
    np.random.seed(0)
    net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 300, verbose=0)
    t = time.perf_counter()
    net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * steps, verbose=0)
    out["indexed_host_planning_only_ms_per_step"] = 1e3 * (time.perf_counter() - t) / steps
    # ... and the generator's index batches alone
    gen = make()
    np.random.seed(0)
    for _ in range(300):
        gen.next_indices()
    t = time.perf_counter()
    rows = 0
    for _ in range(steps):
        rows += len(gen.next_indices()[2])
    out["generator_index_batch_us"] = 1e6 * (time.perf_counter() - t) / steps
    out["mean_rows_per_batch"] = rows / steps
    print(json.dumps(out))


if __name__ == "__main__":
    main()

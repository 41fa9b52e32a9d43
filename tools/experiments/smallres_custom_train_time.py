"""The MTP driver's pre-training (code/ALINK_MTP.py:56-59: SmallRes.customTrainModel on the balanced low-resolution generator):
wall time per generator step (train_on_batch on the kept rows + test_on_batch on the held-out ones), and where it goes."""
import json, os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa: F401
from a_link_amd import siamese, pairs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(3)
people = [rng.randint(0, 256, (2, 64, 64, 3)).astype(np.float32) for _ in range(40)]
low = (32, 32)
model = siamese.SmallRes(low + (3,), (2048,), "probe", 1e-1)
gen = pairs.getGeneratorMTP(pairs.getNormalGenerator(people, 16), 16, resize_res=low)
np.random.seed(0)
model.customTrainModel(gen, 1, 16, 0.2, 16 * 20, verbose=0)           # warm-up: 20 steps
torch.cuda.synchronize()
t = time.perf_counter()
logs = model.customTrainModel(gen, 1, 16, 0.2, 16 * steps, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t
out = {"steps": steps, "ms_per_step": 1e3 * dt / steps, "log": [float(x) for x in logs[0]] if logs else None}
if len(sys.argv) > 2:
    pr = cProfile.Profile()
    pr.enable()
    model.customTrainModel(gen, 1, 16, 0.2, 16 * 100, verbose=0)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print(s.getvalue()[:3500], file=sys.stderr)
print(json.dumps(out))

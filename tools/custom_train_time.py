import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import a_link_amd
from a_link_amd import siamese, pairs
rng = np.random.RandomState(0)
feats = [rng.randn(rng.randint(3, 6), 512).astype(np.float32) for _ in range(200)]
gen = pairs.getGenerator(pairs.getNormalGenerator(feats, 16), pairs.getNormalGenerator(feats, 16), pairs.getImposterGenerator(feats, feats, 16), 16)
net = siamese.SiameseNetwork((512,), "/tmp/ctm", 0.1, seed=1)
np.random.seed(0)
net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 200, verbose=0)
torch.cuda.synchronize()
t = time.perf_counter()
logs = net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 2000, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("customTrainModel: %.3f ms per step (2000 steps), logs %s" % (1e3 * dt / 2000, logs))
# where it goes
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 500, verbose=0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
pstats.Stats(pr).sort_stats("tottime").print_stats(16)

import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import a_link_amd
from a_link_amd import pairs, siamese
rng = np.random.RandomState(0)
feats = [rng.randn(rng.randint(3, 6), 512).astype(np.float32) for _ in range(200)]
def make():
    return pairs.getGenerator(pairs.getNormalGenerator(feats, 16), pairs.getNormalGenerator(feats, 16), pairs.getImposterGenerator(feats, feats, 16), 16)
out = {}
for name in ("null_stream", "side_stream"):
    net = siamese.SiameseNetwork((512,), "/tmp/ctm", 0.1, seed=1)
    gen = make()
    np.random.seed(0)
    ctx = torch.cuda.stream(torch.cuda.Stream()) if name == "side_stream" else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 300, verbose=0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        net.customTrainModel(gen, 1, 16, 0.2, n_steps=16 * 4000, verbose=0)
        torch.cuda.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t) / 4000
print(json.dumps(out))

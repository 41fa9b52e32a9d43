import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import a_link_amd
from a_link_amd import weights as W
from a_link_amd.backbone import IRBackbone
params = W.synthetic_ir_params(W.ARCH_UNITS["r100"], seed=1, normalized=True)
for dt in ("bf16", "f16x2"):
    bb = IRBackbone(params, dtype=dt, max_batch=292, lazy_range_check=True)
    x = torch.randint(0, 256, (1, 112, 112, 3), dtype=torch.uint8).cuda()
    out = torch.empty((1, 512), device="cuda")
    for _ in range(3): bb.embed_device(x, out=out)
    torch.cuda.synchronize()
    ref = out.clone()
    t=time.perf_counter()
    for _ in range(50): bb.embed_device(x, out=out)
    torch.cuda.synchronize(); eager=(time.perf_counter()-t)/50*1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bb.embed_device(x, out=out)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            bb.embed_device(x, out=out)
        g.replay(); torch.cuda.synchronize()
        t=time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); gr=(time.perf_counter()-t)/50*1e3
        print(dt, "eager %.3f ms graph %.3f ms equal %s" % (eager, gr, torch.equal(out, ref)))
    except Exception as e:
        print(dt, "graph capture failed:", str(e)[:300])

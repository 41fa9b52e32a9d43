"""f16x2 mode: are repeated embeddings of the same pool bit-identical, across runs and stream counts? (debug)
usage: x2_determinism.py dtype [linear_mode] [fine_max]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import weights as W, _abi
from a_link_amd.backbone import IRBackbone
dt = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
lib = _abi.load()
if len(sys.argv) > 2 and int(sys.argv[2]) >= 0:
    lib.alink_debug_set_linear(int(sys.argv[2]))
if len(sys.argv) > 3 and int(sys.argv[3]) >= 0:
    lib.alink_debug_set_fine_max(int(sys.argv[3]))
p = W.synthetic_ir_params(W.R50_UNITS, seed=1, normalized=True)
x = torch.randint(0, 256, (2048, 112, 112, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
ref = IRBackbone(p, dtype=dt, max_batch=292, streams=1).embed_device(x).clone()
torch.cuda.synchronize()
for streams in (4,):
    bb = IRBackbone(p, dtype=dt, max_batch=292, streams=streams)
    for rep in range(int(os.environ.get("REPS", "4"))):
        got = bb.embed_device(x)
        torch.cuda.synchronize()
        d = (got - ref).abs().amax(1)
        bad = torch.nonzero(d > 0).flatten().tolist()
        runs, start = [], None
        for i in bad + [None]:
            if start is None:
                start = prev = i
            elif i is not None and i == prev + 1:
                prev = i
            else:
                runs.append((start, prev - start + 1))
                start = prev = i
        print("args %s streams %d rep %d: rows differing %d max %.3e runs(start,len) %s" % (sys.argv[2:], streams, rep, len(bad), float(d.max()) if bad else 0.0, runs[:10]))

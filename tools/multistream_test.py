#!/usr/bin/env python3
"""Does running sub-batches on independent HIP streams (de-synchronising the per-tile HBM bursts) help?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bb = IRBackbone(W.synthetic_ir_params(W.R100_UNITS, seed=1), max_batch=B)
    x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8).float().cuda()
    out = torch.empty((B, 512), device="cuda")
    lib = bb.lib


    def run(nstreams, steps=10):
        sub = B // nstreams
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        wss = []
        for _ in range(nstreams):
            nb = lib.alink_backbone_workspace_bytes(bb.h, sub)
            t = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
            wss.append((t, t.data_ptr() + ((-t.data_ptr()) % 256), nb))

        def once():
            for i, s in enumerate(streams):
                _abi.check(lib.alink_embed(bb.h, C.c_void_p(x[i * sub:(i + 1) * sub].data_ptr()), 0, sub,
                                           C.c_void_p(out[i * sub:(i + 1) * sub].data_ptr()), C.c_void_p(wss[i][1]), wss[i][2],
                                           C.c_void_p(s.cuda_stream)))
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return B / dt


    ref = bb.embed_device(x).clone()
    for ns in (1, 2, 4, 8):
        r = run(ns)
        ok = torch.equal(out, ref)
        print("batch %d streams %d: %.0f emb/s  (identical to single-stream: %s)" % (B, ns, r, ok))


if __name__ == "__main__":
    main()

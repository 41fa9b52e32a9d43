#!/usr/bin/env python3
"""Does it pay to keep the launch chains of different streams at DIFFERENT DEPTHS of the network (one stream in the
HBM-bound front while another is in the MFMA-bound stages)?  n streams loop over forwards of `chunk` images with no
join between forwards; stream i starts i/n of a forward late (offset on) or all together (offset off)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "r100"
    bb = IRBackbone(W.synthetic_ir_params(W.ARCH_UNITS[model], seed=1), max_batch=292, shards_per_call=1)
    lib = bb.lib
    for chunk, nstreams, fine_max in ((292, 4, -1), (146, 8, 0), (146, 4, 0), (292, 2, -1), (292, 8, -1)):
        lib.alink_debug_set_fine_max(384 if fine_max < 0 else fine_max)
        x = torch.randint(0, 256, (chunk, 112, 112, 3), dtype=torch.uint8).float().cuda()
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        outs = [torch.empty((chunk, 512), device="cuda") for _ in range(nstreams)]
        wss = []
        for _ in range(nstreams):
            nb = lib.alink_backbone_workspace_bytes(bb.h, chunk)
            t = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
            wss.append((t, t.data_ptr() + ((-t.data_ptr()) % 256), nb))

        def forward(i):
            _abi.check(lib.alink_embed(bb.h, C.c_void_p(x.data_ptr()), 0, chunk, C.c_void_p(outs[i].data_ptr()),
                                       C.c_void_p(wss[i][1]), wss[i][2], C.c_void_p(streams[i].cuda_stream)))
        for i in range(nstreams):
            forward(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        forward(0)
        torch.cuda.synchronize()
        t_one = time.perf_counter() - t0                      # one forward alone, seconds
        for offset in (False, True):
            reps = 12
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if offset:
                cycles_per_s = 1.0e9                          # torch.cuda._sleep counts ~1 GHz ticks: only the ORDER of magnitude matters
                for i in range(nstreams):
                    with torch.cuda.stream(streams[i]):
                        torch.cuda._sleep(int(i / nstreams * t_one * cycles_per_s))
            for r in range(reps):
                for i in range(nstreams):
                    forward(i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("%s chunk %d x %d streams, offset %s: %.0f embeddings/s" % (model, chunk, nstreams, offset, reps * nstreams * chunk / dt))


if __name__ == "__main__":
    main()

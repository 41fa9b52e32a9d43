#!/usr/bin/env python3
"""Which launch of the chain first differs under concurrency: alink_embed stopped after STOP convolution launches, the whole
workspace of every chunk compared between a serial and a 4-stream issue of the same 7 chunks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import a_link_amd  # noqa
from a_link_amd import _abi, weights as W
from a_link_amd.backbone import IRBackbone
lib = _abi.load()
p = W.synthetic_ir_params((1, 1, 1, 1), seed=1, normalized=True)
x = torch.randint(0, 256, (2048, 112, 112, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
streams, reps, stop = 4, int(os.environ.get("REPS", "20")), int(os.environ.get("STOP", "1"))
lib.alink_debug_set_s2direct(int(os.environ.get("S2", "0")))
bb = IRBackbone(p, dtype="f16", max_batch=292, streams=1, shards_per_call=1, lazy_range_check=True)
lib.alink_debug_set_stop_after(stop)
nb = lib.alink_backbone_workspace_bytes(bb.h, 292)
chunks = [(i, min(292, 2048 - i)) for i in range(0, 2048, 292)]
def run(side):
    wss = [torch.zeros(nb + 256, dtype=torch.uint8, device="cuda") for _ in chunks]
    out = torch.zeros((2048, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for j, (i, m) in enumerate(chunks):
        base = wss[j].data_ptr(); off = (-base) % 256
        st = side[j % len(side)]
        _abi.check(lib.alink_embed(bb.h, _abi.ptr(x[i:i + m]), _abi.LAYOUT_NHWC_U8, m, _abi.ptr(out[i:i + m]), C.c_void_p(base + off), nb,
                                   C.c_void_p(st.cuda_stream)), "alink_embed")
    torch.cuda.synchronize()
    return wss
ref = run([torch.cuda.Stream()])
side = [torch.cuda.Stream() for _ in range(streams)]
for rep in range(reps):
    got = run(side)
    for j, (a, b) in enumerate(zip(ref, got)):
        if not torch.equal(a, b):
            d = (a != b).nonzero().flatten()
            print("stop", stop, "rep", rep, "chunk", j, "bytes differing", d.numel(), "first", d[:6].tolist(), "last", d[-1].item(), flush=True)
            off0 = (-a.data_ptr()) % 256
            big = 292 * 112 * 112 * 64 * 2
            dd = d - off0
            xs = dd[dd < 292 * 56 * 56 * 128]
            if xs.numel():
                pix = torch.unique(xs // 128)
                im, r, px = pix // 3136, (pix % 3136) // 56, pix % 56
                print("   xs: %d bytes in %d pixels; images %s rows %s..%s; pixels (img,row,x):" % (xs.numel(), pix.numel(), torch.unique(im).tolist(), r.min().item(), r.max().item()),
                      [(int(a_), int(b_), int(c_)) for a_, b_, c_ in zip(im[:24], r[:24], px[:24])], flush=True)
                for pp_ in pix[:4].tolist():
                    o = off0 + pp_ * 128
                    print("      pixel", pp_, "ref", a[o:o + 128].view(torch.float16).tolist())
                    print("      pixel", pp_, "got", b[o:o + 128].view(torch.float16).tolist(), flush=True)
            c1 = dd[(dd >= big) & (dd < 2 * big)] - big
            if c1.numel():
                pix = torch.unique(c1 // 128)
                im, r, px = pix // 12544, (pix % 12544) // 112, pix % 112
                print("   conv1 out: %d bytes in %d pixels; images %s rows %s..%s; pixels (img,y,x):" % (c1.numel(), pix.numel(), torch.unique(im).tolist(), r.min().item(), r.max().item()),
                      [(int(a_), int(b_), int(c_)) for a_, b_, c_ in zip(im[:24], r[:24], px[:24])], flush=True)
print("done stop", stop)
lib.alink_debug_set_stop_after(0); lib.alink_debug_set_s2direct(1)

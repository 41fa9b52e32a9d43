#!/usr/bin/env python3
"""Per-launch timing table of one backbone forward (HIP events through alink_embed_profile)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import a_link_amd  # noqa
from a_link_amd import weights as W
from a_link_amd.backbone import IRBackbone

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="r100")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ablate", type=int, default=0)
ap.add_argument("--no-direct", action="store_true")
ap.add_argument("--no-pair", action="store_true")
ap.add_argument("--no-c64", action="store_true", help="A/B: the 112-wide 64->64 layer on the row-aligned tile kernel instead of the rolling-row kernel")
ap.add_argument("--no-fuse-stem", action="store_true", help="A/B: stem and stage1_unit1 conv1 as two launches instead of the fused front kernel")
ap.add_argument("--lib", default="", help="experiments: load this build of libalink_hip.so instead of the package's")
ap.add_argument("--no-s2direct", action="store_true", help="A/B: stage1_unit1's stride-2 conv2 + shortcut on the implicit-GEMM kernel")
ap.add_argument("--no-fuse-sc", action="store_true", help="A/B: projection shortcuts as launches of their own")
ap.add_argument("--linear", type=int, default=-1, help="linear-tile widths: bit0 56, bit1 28, bit2 14, bit3 7 (default: library default)")
a = ap.parse_args()
units = W.ARCH_UNITS[a.model]
from a_link_amd import _abi
if a.lib:
    _abi.LIB_PATH = os.path.abspath(a.lib)
_lib = _abi.load()
_lib.alink_debug_set_ablate(a.ablate)
if a.no_direct:
    _lib.alink_debug_set_direct(0)
if a.no_pair:
    _lib.alink_debug_set_pair(0)
if a.linear >= 0:
    _lib.alink_debug_set_linear(a.linear)
if a.no_fuse_sc:
    _lib.alink_debug_set_fuse_shortcut(0)
if a.no_c64:
    _lib.alink_debug_set_c64(0)
if a.no_fuse_stem:
    _lib.alink_debug_set_fuse_stem(0)
if a.no_s2direct:
    _lib.alink_debug_set_s2direct(0)
bb = IRBackbone(W.synthetic_ir_params(units, seed=1), dtype=a.dtype, max_batch=a.batch)
x = torch.randint(0, 256, (a.batch, 112, 112, 3), dtype=torch.uint8).float().cuda()
for _ in range(2):
    bb.embed_device(x)
profs = [bb.profile(x) for _ in range(a.reps)]
ms = np.median(np.array([[m for _, m, _ in p] for p in profs]), axis=0)
kinds = [k for k, _, _ in profs[0]]
fl = [f for _, _, f in profs[0]]
# group identical (kind, flops) launches
fused_sc = sum(1 for k in kinds if k == 1) == 2 * sum(units)     # projection shortcuts inside the conv2 launch
fused_front = kinds[0] == 1                                       # stem inside the first conv launch (front_c64.hip)
names = [] if fused_front else ["stem"]
for s in range(4):
    for u in range(units[s]):
        names.append("stem+s1u1_conv1" if (fused_front and s == 0 and u == 0) else "s%du%d_conv1" % (s + 1, u + 1))
        if u == 0 and not fused_sc:
            names.append("s%du%d_sc" % (s + 1, u + 1))
        names.append("s%du%d_conv2%s" % (s + 1, u + 1, "+sc" if (u == 0 and fused_sc) else ""))
names += ["fc_splitk", "fc_finish"]
tot = ms.sum()
print("total %.3f ms  -> %.0f emb/s ; conv %.3f ms %.1f TF/s" % (
    tot, a.batch / tot * 1e3, sum(m for m, k in zip(ms, kinds) if k == 1),
    sum(f for f, k in zip(fl, kinds) if k == 1) / sum(m for m, k in zip(ms, kinds) if k == 1) / 1e9))
agg = {}
for n, m, f in zip(names, ms, fl):
    key = n if ("u1_" in n or n in ("stem", "fc_splitk", "fc_finish")) else n.split("u")[0] + "uN_" + n.split("_")[1]
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += m; d[2] += f
print("%-14s %4s %9s %7s %9s %6s" % ("layer", "n", "ms_total", "ms_each", "TF/s", "%time"))
for k, (n, m, f) in agg.items():
    print("%-14s %4d %9.3f %7.3f %9.1f %6.1f" % (k, n, m, m / n, f / m / 1e9 if m > 0 else 0, 100 * m / tot))

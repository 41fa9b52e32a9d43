#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc runs of bench.py (FETCH_SIZE in one pass, WRITE_SIZE in another: the
TCC block cannot hold both) as HBM traffic per kernel instantiation AND launch shape.
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled here.
usage: pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv n_forwards"""
import csv
import sys
from collections import defaultdict

fetch_csv, write_csv, n_forwards = sys.argv[1], sys.argv[2], int(sys.argv[3])
TAGS = ("conv3x3_linear_kernel", "conv3x3_direct_kernel", "conv3x3_c64_kernel", "conv3x3_s2c64_kernel", "front_c64_kernel", "conv_igemm_kernel", "stem_kernel", "stem_x2_kernel",
        "fc_finish_kernel")


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for tag in TAGS:
                if tag in r["Kernel_Name"]:
                    key = (tag, int(r["Grid_Size"]))
                    agg[key][0] += 1
                    agg[key][1] += float(r["Counter_Value"])
    return agg


f = load(fetch_csv, "FETCH_SIZE")
w = load(write_csv, "WRITE_SIZE")
print("kernel,grid_threads,launches_per_forward,fetch_MB_per_launch(x2 corrected),write_MB_per_launch,MB_per_forward")
tot = 0.0
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
    fl, fv = f.get(k, [0, 0.0])
    wl, wv = w.get(k, [0, 0.0])
    n = max(fl, wl)
    fmb, wmb = 2 * fv * 1024 / 1e6 / max(fl, 1), wv * 1024 / 1e6 / max(wl, 1)
    per_fwd = (fmb + wmb) * n / n_forwards
    tot += per_fwd
    print("%s,%d,%.1f,%.1f,%.1f,%.1f" % (k[0], k[1], n / n_forwards, fmb, wmb, per_fwd))
print("total_MB_per_forward,,,,,%.1f" % tot)

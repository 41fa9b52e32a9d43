#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py: HBM traffic per forward of the
backbone kernels.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the
bytes of wide coalesced reads (MI355X_MICROARCH.md §HBM) -> doubled here."""
import csv
import sys
from collections import defaultdict

fetch_csv, write_csv, n_forwards = sys.argv[1], sys.argv[2], int(sys.argv[3])


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            for tag in ("conv3x3_direct_kernel", "conv_igemm_kernel", "stem_kernel", "fc_finish_kernel"):
                if tag in k:
                    agg[tag][0] += 1
                    agg[tag][1] += float(r["Counter_Value"])
    return agg


f = load(fetch_csv, "FETCH_SIZE")
w = load(write_csv, "WRITE_SIZE")
print("kernel,launches_per_forward,fetch_MB_per_forward(x2 corrected),write_MB_per_forward")
tot = 0.0
for k in sorted(set(f) | set(w)):
    fl, fv = f.get(k, [0, 0.0])
    wl, wv = w.get(k, [0, 0.0])
    fmb = 2 * fv * 1024 / 1e6 / n_forwards
    wmb = wv * 1024 / 1e6 / n_forwards
    tot += fmb + wmb
    print("%s,%.1f,%.1f,%.1f" % (k, max(fl, wl) / n_forwards, fmb, wmb))
print("total_MB_per_forward,%.1f" % tot)

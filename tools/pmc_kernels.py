#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc runs of ANY command (FETCH_SIZE in one pass, WRITE_SIZE in another), for
kernels whose name contains one of the given tags.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half
the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): the raw figure AND its double are printed.
usage: pmc_kernels.py fetch_counter_collection.csv write_counter_collection.csv tag [tag ...]"""
import csv
import sys
from collections import defaultdict

fetch_csv, write_csv, tags = sys.argv[1], sys.argv[2], sys.argv[3:]


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for tag in tags:
                if tag in r["Kernel_Name"]:
                    key = (tag, int(r["Grid_Size"]))
                    agg[key][0] += 1
                    agg[key][1] += float(r["Counter_Value"])
    return agg


f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
print("kernel,grid_threads,launches,fetch_MB_per_launch_raw,fetch_MB_per_launch_x2,write_MB_per_launch")
for k in sorted(set(f) | set(w)):
    fl, fv = f.get(k, [0, 0.0])
    wl, wv = w.get(k, [0, 0.0])
    print("%s,%d,%d,%.1f,%.1f,%.1f" % (k[0], k[1], max(fl, wl), fv * 1024 / 1e6 / max(fl, 1), 2 * fv * 1024 / 1e6 / max(fl, 1), wv * 1024 / 1e6 / max(wl, 1)))

#!/usr/bin/env python3
"""Copy a rocprofv3 *_kernel_stats.csv into profiles/ with kernel names truncated to 120 chars."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w", newline="") as g:
    r, w = csv.reader(f), csv.writer(g)
    for row in r:
        row[0] = row[0][:120]
        w.writerow(row)

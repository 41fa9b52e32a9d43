#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; print its per-kernel statistics as the CSV that
`--stats` used to produce (one row per kernel instantiation AND launch shape), names cut to 120 chars.
usage: rocprof_db_stats.py results.db > profiles/<name>_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, count(*), sum(duration), avg(duration), "
                 "min(duration), max(duration) from kernels group by name, grid_x, workgroup_x, lds_size "
                 "order by sum(duration) desc").fetchall()
tot = sum(r[6] for r in rows) or 1
print("Name,GridX,WorkgroupX,LdsBytes,VGPRs,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for name, gx, wx, lds, vg, n, s, a, lo, hi in rows:
    print('"%s",%d,%d,%d,%d,%d,%d,%.1f,%.2f,%d,%d' % (name[:120].replace('"', "'"), gx, wx, lds, vg, n, s, a,
                                                      100.0 * s / tot, lo, hi))

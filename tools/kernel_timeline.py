#!/usr/bin/env python3
"""Print the last N kernel launches of a rocprofv3 (rocpd SQLite) trace as a timeline:
gap to the previous kernel's end, duration, grid, name.   usage: kernel_timeline.py results.db [N]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
rows = rows[-n:]
prev = None
print("%9s %9s %7s %5s  %s" % ("gap_us", "dur_us", "grid", "wg", "kernel"))
for name, s, e, gx, wx in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%9.2f %9.2f %7d %5d  %s" % (gap, (e - s) / 1e3, gx, wx, name[:90]))
    prev = e

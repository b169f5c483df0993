#!/usr/bin/env python
"""Kernel-level occupancy of a rocprofv3 kernel trace (rocpd sqlite): per stream-count window, the fraction of
wall time with >= 1 kernel running, the mean number of concurrently running kernels, and the share of time in
which the dominant GEMM runs.  Usage: trace_overlap.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, name, stream_id from kernels order by start").fetchall()
t0 = rows[0][0]
# split the run into windows of 50 ms and report each
W = 50e6
import collections
wins = collections.defaultdict(list)
for s, e, n, st in rows:
    wins[int((s - t0) // W)].append((s, e, n, st))
print("win  kernels streams  busy%  mean_conc  gemm_busy%")
for w in sorted(wins):
    ev = wins[w]
    pts = []
    for s, e, n, st in ev:
        pts.append((s, 1, 'g' if 'gemm_nt' in n else 'o')); pts.append((e, -1, 'g' if 'gemm_nt' in n else 'o'))
    pts.sort()
    cur = 0; curg = 0; last = pts[0][0]; busy = 0; conc = 0; gb = 0
    for t, d, k in pts:
        dt = t - last
        if cur > 0: busy += dt
        conc += cur * dt
        if curg > 0: gb += dt
        cur += d
        if k == 'g': curg += d
        last = t
    span = pts[-1][0] - pts[0][0]
    print("%3d %8d %7d %6.1f %9.2f %10.1f" % (w, len(ev), len({st for *_, st in ev}), 100 * busy / span, conc / span, 100 * gb / span))

#!/usr/bin/env python
"""Durations of the family walk when it is launched twice back to back (CAFEHIP_K2_REPS=2): the second launch finds its code in the
instruction caches.  `rocprofv3 --kernel-trace -d /tmp/wp -o r -- python tools/ab_one.py cfg2:10000 test1`, then this script on the db."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
by = {}
prev = None
for name, s, e in rows:
    if "k2_prune" in name:
        key = (name[:60], "second" if (prev is not None and prev == name) else "first")
        by.setdefault(key, []).append((e - s) / 1e3)
    prev = name
for k, v in sorted(by.items()):
    v = np.array(v[len(v) // 2:])
    if len(v) > 20:
        print("%-62s %-6s n %5d  median %.2f us  p10 %.2f  p90 %.2f" % (k[0], k[1], len(v), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))

#!/usr/bin/env python
"""Gaps between consecutive kernels of a rocprofv3 rocpd database (end of one -> start of the next), grouped by
the (previous, next) kernel pair.  Usage: rocpd_gaps.py <results.db> [skip_first_n_kernels]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = db.execute("select name, start, end from kernels order by start").fetchall()[skip:]
    agg = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        agg.setdefault((n0[:40], n1[:40]), []).append((s1 - e0) / 1e3)
    for (a, b), g in sorted(agg.items(), key=lambda kv: -len(kv[1])):
        g.sort()
        print("%-42s -> %-42s n=%4d  median %8.2f us  min %8.2f  max %8.2f" % (a, b, len(g), g[len(g) // 2], g[0], g[-1]))


if __name__ == "__main__":
    main()

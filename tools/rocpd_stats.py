#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) result: per-kernel calls / total / average /
min / max duration, like `--stats` CSV output.  Usage: rocpd_stats.py <results.db> [skip_first_n]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
                      "accum_vgpr_count, sgpr_count from kernels order by start").fetchall()
    agg = {}
    for r in rows:
        agg.setdefault(r[0], []).append(r)
    tot = sum(r[2] - r[1] for r in rows)
    print("%-70s %6s %12s %12s %12s %12s %6s  %s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct",
                                                      "grid x wg / lds / vgpr+agpr / sgpr"))
    for name, rs in sorted(agg.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        rs2 = rs[skip:] if len(rs) > skip else rs
        d = [(r[2] - r[1]) / 1e3 for r in rs2]
        r = rs2[-1]
        print("%-70s %6d %12.1f %12.2f %12.2f %12.2f %6.1f  %dx%dx%d x %d / %d / %d+%d / %d" % (
            name[:70], len(d), sum(d), sum(d) / len(d), min(d), max(d), 100.0 * sum(r_[2] - r_[1] for r_ in rs) / tot,
            r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main()

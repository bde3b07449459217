#!/usr/bin/env python
"""Per-DISPATCH sums of the PMC counters of a rocprofv3 rocpd database for kernels matching a pattern, largest first:
`python tools/rocpd_pmc_top.py <db> <name pattern> [n]` -- e.g. the batch-mode launches of a run that also holds the
table's objective evaluations under the same kernel name (the batch launches are the ones with the largest sums)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
    key = "dispatch_id" if "dispatch_id" in cols else ("event_id" if "event_id" in cols else None)
    if key is None:
        print("columns of pmc_events:", cols)
        return
    per = {}
    for d, name, cn, v in db.execute("select %s, name, counter_name, counter_value from pmc_events" % key):
        if pat in name:
            per.setdefault(cn, {}).setdefault(d, 0.0)
            per[cn][d] += v
    for cn, dd in sorted(per.items()):
        top = sorted(dd.values(), reverse=True)[:n]
        print("%-28s dispatches %5d   mean of the %d largest per-dispatch sums %16.1f   (median dispatch %14.1f)"
              % (cn, len(dd), len(top), sum(top) / len(top), sorted(dd.values())[len(dd) // 2]))


if __name__ == "__main__":
    main()

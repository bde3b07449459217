#!/usr/bin/env python
"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database.
FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them (derived from the L2's memory-side
request counters); per /opt/skills/guides/MI355X_MICROARCH.md on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads, so the x2-corrected figure is printed next to the raw one."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, counter_name, counter_value from pmc_events").fetchall()
    agg = {}
    for name, cn, v in rows:
        agg.setdefault((name, cn), []).append(v)
    for (name, cn), vs in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(vs) / len(vs)
        extra = "  (x2 gfx950 correction: %.1f KiB)" % (2 * avg) if cn == "FETCH_SIZE" else ""
        unit = "KiB per launch" if cn in ("FETCH_SIZE", "WRITE_SIZE") else "(avg per counter instance record)"
        print("%-60s %-24s records %5d  avg %12.1f %s%s" % (name[:60], cn, len(vs), avg, unit, extra))


if __name__ == "__main__":
    main()

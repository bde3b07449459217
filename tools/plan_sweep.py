"""Evaluation time of a table under variations of the compression plan (round 5): which levels of the
compressed forest pay for their launch.

    python tools/plan_sweep.py test1|cfg2|cfg3|cfg4|cfg5|turnover [key=v1,v2,... ...]   e.g.  compress_max_level=0,1,2,3,4

Every combination of the listed option values: ms per evaluation (median of 3 x 200 steps after priming), the plan the
library chose (levels, tiles per level, walk steps)."""
import itertools
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def workload(name):
    if name == "test1":
        return bench.test1_workload()
    if name == "turnover":
        return bench.turnover_workload("cfg2")
    return bench.synthetic_workload(name, 0, 1, "weak", None)


def measure(w, options, steps=200):
    import cafe_amd
    eng = cafe_amd.Engine(0)
    for k, v in options.items():
        eng.set_option(k, v)
    leg = bench.Leg(w, 0, None, shared_engine=eng)
    leg.prepare_rates(steps + 8)
    leg.prime()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for s in range(steps):
            leg.step(s)
        ts.append((time.perf_counter() - t0) / steps)
    desc = eng.describe()
    eng.close()
    return sorted(ts)[1], desc


def main():
    name = sys.argv[1]
    grids = []
    for a in sys.argv[2:]:
        k, vs = a.split("=", 1)
        grids.append([(k, v) for v in vs.split(",")])
    w = workload(name)
    for combo in itertools.product(*grids) if grids else [()]:
        opts = dict(combo)
        ms, desc = measure(w, opts)
        m = re.search(r"compressed\(([^)]*)\)", desc)
        print("%-10s %-40s %8.2f us   %s" % (name, " ".join("%s=%s" % kv for kv in combo), 1e6 * ms, m.group(1) if m else "uncompressed"), flush=True)


if __name__ == "__main__":
    main()

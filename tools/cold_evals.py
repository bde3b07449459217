#!/usr/bin/env python
"""Where the first search of a process loses its time (round 6): wall clock of each of the first N
objective evaluations of a FRESH process on the configs[1] table (or test1), one line per evaluation group.

    python tools/cold_evals.py [cfg2:10000|test1] [k=v ...]      (options are handed to the engine before the table)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    t_start = time.perf_counter()
    spec = sys.argv[1] if len(sys.argv) > 1 else "cfg2:10000"
    import cafe_amd
    from k2c_ab import workload
    tree, counts, cfg, rng, prior, nl, nm = workload(spec)
    t0 = time.perf_counter()
    eng = cafe_amd.Engine(0)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        eng.set_option(k, v)
    t1 = time.perf_counter()
    tree.apply(eng)
    eng.set_families(counts, rng)
    t2 = time.perf_counter()
    ts = []
    for i in range(120):
        a = time.perf_counter()
        eng.get_posterior(nl * (1.0 + 0.001 * i), nm, prior)
        ts.append(time.perf_counter() - a)
    ts = 1e6 * np.array(ts)
    print("%s: context %.1f ms, tree + table %.1f ms; evaluation wall clock in us:" % (spec, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print("   first 10: " + " ".join("%.0f" % x for x in ts[:10]))
    for lo in range(10, 120, 10):
        print("   %3d-%3d: mean %.1f  (%s)" % (lo, lo + 9, ts[lo:lo + 10].mean(), " ".join("%.0f" % x for x in ts[lo:lo + 10])))
    print("   sum of the first 59 evaluations %.3f ms; 59 x the steady mean (last 30) %.3f ms" % (ts[:59].sum() / 1e3, 59 * ts[-30:].mean() / 1e3))
    print("   " + eng.describe()[-200:])
    eng.close()


if __name__ == "__main__":
    main()

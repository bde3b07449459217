#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE of the family walk of one workload (tools/ab_one.py) under the current environment's options.
python tools/pmc_walk.py cfg3:100000 [label]"""
import glob, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "/tmp/pmc_walk_" + counter
    subprocess.call(["rm", "-rf", d])
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--", sys.executable, "tools/ab_one.py", sys.argv[1]],
                          cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
    agg = {}
    for n, c, v in db.execute("select name, counter_name, counter_value from pmc_events"):
        if c == counter and "k2_prune" in n:
            agg.setdefault(n, []).append(v)
    vals = max(agg.values(), key=len)
    vals = vals[len(vals) // 2:]
    res[counter] = sum(vals) / len(vals)
print("%s %s: FETCH_SIZE %.1f KiB x2 + WRITE_SIZE %.1f KiB = %.3f GB per walk launch" % (sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", res["FETCH_SIZE"], res["WRITE_SIZE"],
      (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024 / 1e9))

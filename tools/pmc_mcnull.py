#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE of the Monte-Carlo-null launch (tools/mcnull_one.py) under the current environment's options:
prints bytes per launch (largest k2_prune launches of the run).  python tools/pmc_mcnull.py [label]"""
import glob, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "/tmp/pmc_mcnull_" + counter
    subprocess.call(["rm", "-rf", d])
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--", sys.executable, "tools/mcnull_one.py", "4"],
                          cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
    vals = sorted(v for n, c, v in db.execute("select name, counter_name, counter_value from pmc_events") if c == counter and "k2_prune" in n)
    res[counter] = sum(vals[-4:]) / 4
print("%s: FETCH_SIZE %.1f KiB x2 + WRITE_SIZE %.1f KiB = %.3f GB per launch" % (sys.argv[1] if len(sys.argv) > 1 else "", res["FETCH_SIZE"], res["WRITE_SIZE"],
      (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024 / 1e9))

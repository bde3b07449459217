#!/usr/bin/env python
"""HBM-side traffic of the dominant kernel of every bench workload, for bench.py's roofline.traffic:
two rocprofv3 passes per workload (FETCH_SIZE and WRITE_SIZE cannot share a pass), each in its own run with
--kernel-trace only (the PMC guidance of /opt/skills/guides/MI355X_MICROARCH.md).  On gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes: it is doubled; WRITE_SIZE is taken as reported -- both factors measured on known-bytes
kernels with this walk's access patterns (profiles/r03_fetch_size_calibration.txt: x2.000 / x1.000).  Writes
<out>/<ROUND_TAG, default r06>_pmc_traffic.json (keys "<config>:<families per launch>:<k2|mcnull>") and a text summary.

    python tools/collect_pmc.py [out_dir]          (on the GPU box; needs rocprofv3)"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = [("cfg2", 10000, "k2"), ("cfg3", 100000, "k2"), ("cfg4", 62500, "k2"), ("cfg5", 100000, "k2"), ("cfg5", 250000, "mcnull")]


def run_pass(counter, cmd, out_dir, tag, largest=0):
    d = os.path.join(out_dir, "%s_%s" % (tag, counter))
    subprocess.call(["rm", "-rf", d])
    full = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--"] + cmd
    subprocess.check_call(full, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        raise RuntimeError("no rocpd database under " + d)
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select name, counter_name, counter_value from pmc_events").fetchall()
    agg, tables = {}, []
    for name, cn, v in rows:
        if cn != counter:
            continue
        if "k2_prune" in name:
            agg.setdefault(name, []).append(v)
        elif "k2c_nodes" in name or "k2c_gemm" in name:
            tables.append(v)
    # the dominant kernel = the instantiation launched most often (the tuner's pick; the timed launches)
    name = max(agg, key=lambda k: len(agg[k]))
    vals = agg[name]
    if largest:
        # the Monte-Carlo-null run: the table's own evaluations (45, to settle the wave grid) use the same
        # instantiation; the `largest` null launches (250 k uncompressed rows each) are the ones with the most traffic
        allv = sorted(v for vs in agg.values() for v in vs)
        tail = allv[-largest:]
        return "batch-mode launch (largest %d of %d k2_prune launches)" % (largest, len(allv)), sum(tail) / len(tail), largest, 0.0
    tail = vals[len(vals) // 2:]          # steady state: the second half of its launches
    per_launch = sum(tail) / len(tail)
    # the factor tables of compressed subtrees (k2c_nodes, one launch per level before EVERY walk launch of whatever
    # instantiation): their mean per evaluation is added to the walk's
    n_walks = sum(len(v) for v in agg.values())
    per_eval_tables = sum(tables) / n_walks if tables and n_walks else 0.0
    return name, per_launch, len(vals), per_eval_tables


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
    os.makedirs(out_dir, exist_ok=True)
    res, lines = {}, []
    for cfg, F, kind in WORK:
        cmd = ([sys.executable, "tools/mcnull_one.py", "6"] if kind == "mcnull" else
               [sys.executable, "tools/ab_one.py", "%s:%d" % (cfg, F)])
        tag = "%s_%d_%s" % (cfg, F, kind)
        kname, fetch_kib, n, t_fetch = run_pass("FETCH_SIZE", cmd, out_dir, tag, 6 if kind == "mcnull" else 0)
        kname2, write_kib, n2, t_write = run_pass("WRITE_SIZE", cmd, out_dir, tag, 6 if kind == "mcnull" else 0)
        traffic = (2.0 * fetch_kib + write_kib) * 1024.0
        tables = (2.0 * t_fetch + t_write) * 1024.0
        res["%s:%d:%s" % (cfg, F, kind)] = {
            "kernel": kname, "launches_seen": n, "fetch_kib_raw": fetch_kib, "fetch_kib_x2": 2.0 * fetch_kib,
            "write_kib": write_kib, "traffic_bytes": traffic, "fetch_factor": 2,
            "tables_traffic_bytes": tables, "tables_fetch_kib_raw": t_fetch, "tables_write_kib": t_write,
            "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate runs), FETCH_SIZE x2 / WRITE_SIZE x1 (calibrated: profiles/r03_fetch_size_calibration.txt), "
                    "mean over the second half of the kernel's launches; tables_* = the k2c_gemm / k2c_nodes launches of one "
                    "evaluation (factor tables of compressed subtrees), 0 where the table does not compress",
        }
        lines.append("%-22s %-60s launches %4d  FETCH_SIZE %12.1f KiB (x2 = %12.1f)  WRITE_SIZE %12.1f KiB  -> %.3f MB per launch"
                     "   + factor tables per evaluation: FETCH %.1f KiB (x2) WRITE %.1f KiB -> %.3f MB"
                     % ("%s:%d:%s" % (cfg, F, kind), kname[:60], n, fetch_kib, 2 * fetch_kib, write_kib, traffic / 1e6,
                        t_fetch, t_write, tables / 1e6))
        print(lines[-1], flush=True)
    tag = os.environ.get("ROUND_TAG", "r06")
    json.dump(res, open(os.path.join(out_dir, tag + "_pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(out_dir, tag + "_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

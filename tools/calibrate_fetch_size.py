#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the pruning kernels (on the GPU box):
builds tools/fetch_size_probe.hip, runs it under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes,
--kernel-trace only) and prints known bytes / reported KiB per kernel.  Writes <out>/r03_fetch_size_calibration.{txt,json}.

    python tools/calibrate_fetch_size.py [out_dir]"""
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "calib")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "fetch_size_probe")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "fetch_size_probe.hip")])
    known = None
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out_dir, "pmc_" + counter)
        subprocess.call(["rm", "-rf", d])
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--", exe], capture_output=True, text=True)
        m = re.search(r"bytes stream16 (\d+) stream8 (\d+) b_operand (\d+) col_gather (\d+) store8 (\d+)", p.stdout)
        known = dict(zip(("stream16", "stream8", "b_operand", "col_gather", "store8"), map(int, m.groups())))
        db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
        for name, cn, v in db.execute("select name, counter_name, counter_value from pmc_events").fetchall():
            if cn != counter:
                continue
            for k in known:
                if k in name:
                    res.setdefault(k, {})[counter] = res.get(k, {}).get(counter, 0.0) + v
    lines = ["kernel        known_bytes      FETCH_SIZE_KiB   bytes/(KiB*1024)    WRITE_SIZE_KiB   bytes/(KiB*1024)"]
    out = {}
    for k, b in known.items():
        f, w = res.get(k, {}).get("FETCH_SIZE", 0.0), res.get(k, {}).get("WRITE_SIZE", 0.0)
        ff = b / (f * 1024.0) if f > 0 else None
        wf = b / (w * 1024.0) if w > 0 else None
        out[k] = {"known_bytes": b, "fetch_kib": f, "write_kib": w, "fetch_factor": ff, "write_factor": wf}
        lines.append("%-12s %12d %16.1f %18s %16.1f %18s" % (k, b, f, "%.3f" % ff if ff else "-", w, "%.3f" % wf if wf else "-"))
    lines.append("(a read kernel's WRITE_SIZE and store8's FETCH_SIZE are incidental traffic; the factor that matters is the"
                 " one of the kernel's own direction)")
    txt = "\n".join(lines) + "\n"
    print(txt)
    open(os.path.join(out_dir, "r03_fetch_size_calibration.txt"), "w").write(txt)
    json.dump(out, open(os.path.join(out_dir, "r03_fetch_size_calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

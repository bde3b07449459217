#!/bin/bash
# round 3, call ac: co-resident workgroups of the cfg2 walk started out of step (option k2stagger, 100 MHz ticks)
mkdir -p gpurun_out/r03ac; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ac
run() { (timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b.json 2> $O/b.err)
  python - <<PY
import json, re
try:
    d=json.load(open("$O/b.json")); r=d["roofline"]
    m=re.search(r"k2:\S+.*?grid=\d+", d.get("engine",""))
    print("$1 ms_per_step %.4f walk %.4f  %s" % (d["ms_per_step"], r["avg_launch_ms"], m.group(0) if m else ""))
except Exception as e: print("$1 failed", e, open("$O/b.err").read()[-300:])
PY
}
unset CAFEHIP_MFMA CAFEHIP_K2CFG CAFEHIP_K2CFG4 CAFEHIP_K2STAGGER; run "default"
for g in 5,3,1,4 4,3,1,4 3,3,1,4; do for st in 0 200 400 600 900; do export CAFEHIP_MFMA=4 CAFEHIP_K2CFG4=$g CAFEHIP_K2STAGGER=$st; run "4x4 $g stagger $st"; done; done
unset CAFEHIP_K2CFG4
for g in 1,3,1,4; do for st in 0 300 600; do export CAFEHIP_MFMA=16 CAFEHIP_K2CFG=$g CAFEHIP_K2STAGGER=$st; run "16x16 $g stagger $st"; done; done

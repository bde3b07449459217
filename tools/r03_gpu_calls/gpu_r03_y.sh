#!/bin/bash
# round 3, call y: 16x16x4 wave grids pinned on the cfg2 walk (the measured choice is the 4x4x4 kernel at 5,3,2,4)
mkdir -p gpurun_out/r03y; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03y
for g in default 1,3,1,4 1,3,2,4 2,3,1,4 1,5,1,2 1,5,2,2 2,5,1,2 1,2,1,5 2,2,1,5 1,4,1,3 1,4,2,3 2,4,1,3; do
  if [ $g = default ]; then unset CAFEHIP_MFMA CAFEHIP_K2CFG; else export CAFEHIP_MFMA=16 CAFEHIP_K2CFG=$g; fi
  (timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b.json 2> $O/b.err)
  python - <<PY
import json, re
try:
    d=json.load(open("$O/b.json")); r=d["roofline"]
    m=re.search(r"k2:\S+.*?grid=\d+", d.get("engine",""))
    print("k2cfg $g ms_per_step %.4f walk %.4f  %s" % (d["ms_per_step"], r["avg_launch_ms"], m.group(0) if m else ""))
except Exception as e: print("k2cfg $g failed", e, open("$O/b.err").read()[-300:])
PY
done

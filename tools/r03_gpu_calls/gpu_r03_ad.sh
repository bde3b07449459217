#!/bin/bash
# round 3, call ad: which workgroups share a CU?  stagger by (blockIdx / div) & 1 for several div
mkdir -p gpurun_out/r03ad; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ad
run() { (timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b.json 2> $O/b.err)
  python - <<PY
import json, re
try:
    d=json.load(open("$O/b.json")); r=d["roofline"]
    print("$1 ms_per_step %.4f walk %.4f" % (d["ms_per_step"], r["avg_launch_ms"]))
except Exception as e: print("$1 failed", e, open("$O/b.err").read()[-300:])
PY
}
export CAFEHIP_MFMA=4 CAFEHIP_K2CFG4=5,3,1,4
for div in 1 2 4 8 16 32 64 128; do for st in 400; do export CAFEHIP_K2STAGGER=$st CAFEHIP_K2STAGGER_DIV=$div; run "4x4 5,3,1,4 stagger $st div $div"; done; done

#!/bin/bash
# round 3, call r: cfg2 walk -- what the grid measurement saw, and pinned small-tile grids
mkdir -p gpurun_out/r03r; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03r
(timeout 600 python tools/mcnull_one.py 5 > $O/mcnull.log 2>&1); grep "^mcnull" $O/mcnull.log | cut -c1-330
(CAFEHIP_K2TUNE_LOG=1 timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/bench_tunelog.json 2> $O/bench_tunelog.err); grep -i "tune\|cand\|grid" $O/bench_tunelog.err | head -40
for g in 5,4,1,4 5,2,1,8 3,4,1,4 3,4,2,4 4,4,2,4 2,4,2,4 5,3,2,4 5,4,2,2 6,3,1,4 7,2,1,4 5,2,2,4; do
  (CAFEHIP_MFMA=4 CAFEHIP_K2CFG4=$g timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b.json 2> $O/b.err)
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); r=d["roofline"]
    print("k2cfg4 $g ms_per_step %.4f walk %.4f" % (d["ms_per_step"], r.get("kernel_ms_events", r.get("kernel_ms", 0)) or 0), d["config"].get("describe","")[-200:])
except Exception as e: print("k2cfg4 $g failed", e)
PY
done

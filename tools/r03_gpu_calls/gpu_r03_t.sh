#!/bin/bash
# round 3, call t: does the objective walk pay for the per-count product instantiations? same box, three builds, cfg3 + cfg2 + null
mkdir -p gpurun_out/r03t; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03t
for rep in 1 2; do
for v in default nofew deep; do
  if [ $v = default ]; then unset CAFEHIP_LIB; else export CAFEHIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$v/libcafehip.so; fi
  for c in cfg3 cfg2; do
  (timeout 600 python bench.py --config $c --steps 120 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b.json 2> $O/b.err)
  python - <<PY
import json
d=json.load(open("$O/b.json")); r=d["roofline"]
print("$v $c rep $rep ms_per_step %.4f walk %.4f" % (d["ms_per_step"], r["avg_launch_ms"]))
PY
  done
  (timeout 600 python tools/mcnull_one.py 5 > $O/m.log 2>&1); grep "^mcnull" $O/m.log | cut -c1-100 | sed "s/^/$v /"
done
done

#!/bin/bash
# round 3, call q: cfg2 -- walk stamps, compression threshold sweep with this round's table kernel; null launch with the batch grid rule
mkdir -p gpurun_out/r03q; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03q
(timeout 600 python tools/mcnull_one.py 5 > $O/mcnull.log 2>&1); grep "^mcnull" $O/mcnull.log | cut -c1-400
(timeout 600 python -m pytest tests/test_gpu_batch_trim.py -m gpu -q > $O/pytest.log 2>&1); tail -2 $O/pytest.log
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 600 python tools/k2_stamps.py cfg2 > $O/stamps_cfg2.log 2>&1); tail -30 $O/stamps_cfg2.log | cut -c1-200
for th in default 0.02 0.05 0.1 0.2 0.35 0.5 0.8; do
  if [ $th = default ]; then unset CAFEHIP_COMPRESS_THETA; else export CAFEHIP_COMPRESS_THETA=$th; fi
  (timeout 600 python bench.py --steps 300 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/bench_theta_$th.json 2> $O/bench_theta_$th.err)
  python - <<PY
import json
d=json.load(open("$O/bench_theta_$th.json"))
r=d["roofline"]
print("theta $th ms_per_step %.4f walk_ms %s tables %s" % (d["ms_per_step"], r.get("kernel_ms"), json.dumps(r.get("factor_tables"))[:200]))
PY
done

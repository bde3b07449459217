#!/bin/bash
# round 3, call l: the >127-taxa fuzz cases and the generator-prior search leg (cfg3, cfg5)
mkdir -p gpurun_out/r03l; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03l; R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/pytest_fuzz.log 2>&1; echo "rc=$?" >> $O/pytest_fuzz.log); tail -6 $O/pytest_fuzz.log | cut -c1-200
for c in cfg3 cfg5; do
(timeout 900 python bench.py --config $c --no-cpu-baseline --no-probes --no-strong --no-tables > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err)
python - <<PY
import json
d=json.load(open("$O/bench_$c.json"))
for k,v in d["lambda_search"].items():
    if isinstance(v,dict): print("$c",k,v["wall_s"],v["evaluations"],v["fitted"],v["simulated_rates"],v["score"])
PY
done

#!/bin/bash
# round 3, call u: bench.py's native -> torch fallback (forced), one rank and two ranks on one device; the bench contract tests
mkdir -p gpurun_out/r03u; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03u
(timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -4 $O/pytest.log | cut -c1-300
(BENCH_FORCE_NATIVE_FAILURE=1 CAFEHIP_COMM_TIMEOUT_S=8 timeout 600 python bench.py --gpus 2 --same-device --steps 30 --no-cpu-baseline --no-search --no-probes > $O/bench_2rank_fallback.json 2> $O/bench_2rank_fallback.err; echo "rc=$?" >> $O/bench_2rank_fallback.err); tail -3 $O/bench_2rank_fallback.err | cut -c1-300
python - <<PY
import json
d=json.load(open("$O/bench_2rank_fallback.json"))
print(d["comm"], d.get("comm_fallback"), d["ms_per_step"], d.get("strong_scaling",{}).get("ms_per_step"), d["exchange"])
PY
(timeout 600 python bench.py --gpus 2 --same-device --steps 30 --no-cpu-baseline --no-search --no-probes > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?" >> $O/bench_2rank.err); tail -1 $O/bench_2rank.err
python - <<PY
import json
d=json.load(open("$O/bench_2rank.json"))
print(d["comm"], d.get("comm_fallback"), d["ms_per_step"], d.get("strong_scaling",{}).get("ms_per_step"), json.dumps(d["exchange"])[:300])
PY

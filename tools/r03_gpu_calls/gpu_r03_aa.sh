#!/bin/bash
# round 3, call aa: set-up after the threaded compression plan and the open-addressing row dedup
mkdir -p gpurun_out/r03aa; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03aa
(timeout 1500 python -m pytest tests/test_gpu_compression.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_host_driver.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -4 $O/pytest.log | cut -c1-300
for c in cfg2 cfg3 cfg4; do
(timeout 600 python bench.py --config $c --steps 60 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b_$c.json 2> $O/b_$c.err)
python - <<PY
import json
d=json.load(open("$O/b_$c.json"))
print("$c", d["ms_per_step"], d["config"]["last_score"], json.dumps(d["setup_ms"]["set_families_detail_ms"]))
PY
done

#!/bin/bash
# round 3, call ae: k2c_nodes with its set-up in three load stages
mkdir -p gpurun_out/r03ae; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ae; R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_compression.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_multi_eval.py tests/test_gpu_cluster.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log | cut -c1-300
for c in cfg2 cfg3 cfg4; do
(timeout 600 python bench.py --config $c --steps 200 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/b_$c.json 2> $O/b_$c.err)
python - <<PY
import json
d=json.load(open("$O/b_$c.json")); r=d["roofline"]
print("$c ms_per_step %.4f walk %.4f tables %.4f (frac %.3f)" % (d["ms_per_step"], r["avg_launch_ms"], r["factor_tables"]["ms_per_evaluation"], r["factor_tables"]["frac"]))
PY
done
cd /tmp && rm -rf /tmp/kt && (timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/kt_bench_line.json 2>$O/kt.err); cd $R
python tools/step_timeline.py $(find /tmp/kt -name "*.db" | head -1) 150 > $O/timeline_cfg2.txt 2>&1; cat $O/timeline_cfg2.txt

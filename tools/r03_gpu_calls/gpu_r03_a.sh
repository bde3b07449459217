#!/bin/bash
# round 3, call a: baseline timeline of a cfg2 evaluation (where do the 21 us between launches go?)
mkdir -p gpurun_out/r03a; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03a; R=$GRAFT_REPO_ROOT
(timeout 600 python bench.py --no-cpu-baseline --no-search > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?" >> $O/bench_cfg2.err)
cd /tmp && (timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --steps 400 --no-cpu-baseline --no-search --no-probes > $O/kt_bench.json 2>$O/kt.err); cd $R
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/step_timeline.py $DB 300 > $O/timeline_cfg2.txt 2>&1
python tools/rocpd_stats.py $DB > $O/kernel_stats_cfg2.txt 2>&1
cat $O/timeline_cfg2.txt; head -c 600 $O/bench_cfg2.json; echo; head -c 400 $O/kt_bench.json; echo
# native-comm world-1 tests (existing)
(timeout 900 python -m pytest tests/test_gpu_native_comm.py -q > $O/native.log 2>&1; tail -3 $O/native.log)

#!/bin/bash
# round 3, call b: TU split + option API + k2c prefetch: full GPU suite, bench, timeline (A/B k2c_prefetch)
mkdir -p gpurun_out/r03b; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03b; R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log); tail -2 $O/smoke.log
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -15 $O/pytest_gpu.log
for pf in 1 0; do
  (CAFEHIP_K2C_PREFETCH=$pf timeout 600 python bench.py --no-cpu-baseline --no-search --no-probes > $O/bench_cfg2_pf$pf.json 2> $O/bench_cfg2_pf$pf.err; echo "rc=$?" >> $O/bench_cfg2_pf$pf.err)
  head -c 400 $O/bench_cfg2_pf$pf.json; echo
done
cd /tmp && (timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --steps 400 --no-cpu-baseline --no-search --no-probes > $O/kt_bench.json 2>$O/kt.err); cd $R
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/step_timeline.py $DB 300 > $O/timeline_cfg2.txt 2>&1; cat $O/timeline_cfg2.txt
for c in cfg3 cfg4; do
cd /tmp && rm -rf /tmp/kt2 && (timeout 600 rocprofv3 --kernel-trace -d /tmp/kt2 -o r -- python $R/bench.py --config $c --steps 60 --no-cpu-baseline --no-search --no-probes > $O/kt_bench_$c.json 2>$O/kt_$c.err); cd $R
python tools/step_timeline.py $(find /tmp/kt2 -name "*.db" | head -1) 40 > $O/timeline_$c.txt 2>&1; cat $O/timeline_$c.txt
done

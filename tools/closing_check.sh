#!/bin/bash
# After a change that touches only the table kernel of the large tables: the GPU suite, smoke, the bench lines of every
# configuration, rocprofv3 kernel stats + timelines of cfg3 / cfg4 and the PMC traffic passes again (a subset of tools/final_pass.sh;
# everything lands under gpurun_out/final/).
mkdir -p gpurun_out/final; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/final; R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log); tail -2 $O/smoke.log
(timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -4 $O/pytest_gpu.log | cut -c1-200
(timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?" >> $O/bench_cfg2.err); head -c 200 $O/bench_cfg2.json; echo
for c in cfg3 cfg4 cfg5; do
  (timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err); head -c 200 $O/bench_$c.json; echo
done
for c in cfg3 cfg4; do
cd /tmp && rm -rf /tmp/kt2 && (timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o r -- python $R/bench.py --config $c --steps 60 --no-cpu-baseline --no-search --no-probes > $O/kt_bench_line_$c.json 2>$O/kt_$c.err); cd $R
python tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1; python tools/chain_timeline.py $(find /tmp/kt2 -name "*.db" | head -1) 400 > $O/timeline_$c.txt 2>&1
done
(timeout 2400 python tools/collect_pmc.py $O/pmc > $O/pmc.log 2>&1; echo "rc=$?" >> $O/pmc.log); tail -7 $O/pmc.log | cut -c1-300

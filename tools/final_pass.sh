#!/bin/bash
# The round's closing measurements on one MI355X box (everything lands under gpurun_out/final/; copy what is to be judged into
# profiles/): smoke, the whole GPU suite, the default bench line (every single-GPU configuration is a leg of it since round 6) and
# the per-config lines, the N > 1 command as the driver starts it (one rank forced through the exchange; two ranks sharing the
# box's GPU), rocprofv3 kernel stats + one-evaluation timeline of the default bench command and of cfg3 / cfg4, PMC traffic of
# every workload (separate --pmc passes), the strong-scaling blocks, the first evaluations of a fresh process.
#   gpurun --timeout 3600 -- 'bash tools/final_pass.sh [quick]'
mkdir -p gpurun_out/final; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/final; R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log); tail -2 $O/smoke.log
if [ "$1" != "quick" ]; then
(timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -6 $O/pytest_gpu.log | cut -c1-200
fi
(timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_driver_sized.json 2> $O/bench_cfg2_driver.err; echo "rc=$?" >> $O/bench_cfg2_driver.err); head -c 300 $O/bench_cfg2_driver_sized.json; echo
(timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?" >> $O/bench_cfg2.err); head -c 300 $O/bench_cfg2.json; echo
for c in cfg3 cfg4 cfg5; do
  (timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err); head -c 300 $O/bench_$c.json; echo
done
(timeout 600 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-search --no-probes --no-configs > $O/bench_cfg2_one_rank_native_exchange.json 2> $O/bench_forcedist.err; echo "rc=$?" >> $O/bench_forcedist.err)
(timeout 900 python bench.py --gpus 2 --same-device --no-cpu-baseline --no-search --no-probes > $O/bench_cfg2_2rank_same_device.json 2> $O/bench_2rank.err; echo "rc=$?" >> $O/bench_2rank.err)
cd /tmp && rm -rf /tmp/kt && (timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-search --no-probes --no-strong --no-tables --no-configs > $O/kernel_stats_cfg2_bench_line.json 2>$O/kt.err); cd $R
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/kernel_stats_cfg2.txt 2>&1; python tools/chain_timeline.py $DB 2000 > $O/timeline_cfg2.txt 2>&1; cat $O/timeline_cfg2.txt | head -14
for c in cfg3 cfg4; do
cd /tmp && rm -rf /tmp/kt2 && (timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o r -- python $R/bench.py --config $c --steps 60 --no-cpu-baseline --no-search --no-probes > $O/kt_bench_line_$c.json 2>$O/kt_$c.err); cd $R
python tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1; python tools/chain_timeline.py $(find /tmp/kt2 -name "*.db" | head -1) 400 > $O/timeline_$c.txt 2>&1
done
(CAFEHOST_TIMING=1 timeout 600 python tools/cfg5_pipeline_time.py 100000 > $O/cfg5_pipeline.txt 2>&1)
(ROUND_TAG=r06 timeout 2400 python tools/collect_pmc.py $O/pmc > $O/pmc.log 2>&1; echo "rc=$?" >> $O/pmc.log); tail -7 $O/pmc.log | cut -c1-300
(timeout 900 python tools/lookahead_ab.py cfg2 test1 --reps 4 2>&1 | grep -v WARNING > $O/lookahead_ab.txt); cat $O/lookahead_ab.txt
for t in cfg2:10000 test1; do timeout 300 python tools/cold_evals.py $t >> $O/cold_evaluations.txt 2>&1; done
if [ "$1" != "quick" ]; then
(timeout 1500 python tools/strong_blocks.py > $O/strong_blocks.txt 2>&1)
(timeout 1500 tools/first_node.sh --same-device > $O/first_node_dry_run.txt 2>&1); tail -12 $O/first_node_dry_run.txt
fi

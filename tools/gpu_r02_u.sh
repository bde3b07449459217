#!/bin/bash
# full validation + measurement after the K2 inner-loop rework and subtree-state compression
mkdir -p gpurun_out/r02u; export TMPDIR=/tmp; O=gpurun_out/r02u; R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log)
(timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
for c in cfg2 cfg3 cfg4 cfg5; do
  (timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err)
done
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-search --no-probes > /tmp/kt_bench.json 2>/tmp/kt.err); cd $R
python tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > $O/kernel_stats_cfg2.txt 2>&1; cp /tmp/kt_bench.json $O/kernel_stats_cfg2_bench_line.json
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o r -- python $R/bench.py --config cfg3 --steps 40 --no-cpu-baseline --no-search --no-probes > /tmp/kt3_bench.json 2>/tmp/kt3.err); cd $R
python tools/rocpd_stats.py $(find /tmp/kt3 -name "*.db" | head -1) > $O/kernel_stats_cfg3.txt 2>&1; cp /tmp/kt3_bench.json $O/kernel_stats_cfg3_bench_line.json
(timeout 1500 python tools/collect_pmc.py $O/pmc > $O/pmc.log 2>&1; echo "rc=$?" >> $O/pmc.log)
tail -n 3 $O/smoke.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -B5 "Error\|FAILED" $O/pytest_gpu.log | head -40
for c in cfg2 cfg3 cfg4 cfg5; do echo "== $c"; head -c 1500 $O/bench_$c.json; echo; tail -n 2 $O/bench_$c.err | cut -c1-200; done
head -12 $O/kernel_stats_cfg2.txt | cut -c1-200; tail -5 $O/pmc.log
# the same workloads without compression (A/B), and the threshold sweep
(for s in 1 0; do echo "== CAFEHIP_COMPRESS=$s"; CAFEHIP_COMPRESS=$s timeout 900 python tools/ab_one.py cfg2:10000 cfg3:100000 cfg4:62500 cfg5:100000 2>&1 | grep "^cfg" | cut -c1-400; done) > $O/compression_ab.txt 2>&1
(CAFEHIP_K2CFG4=5,3,2,4 CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg2 2>&1 | grep -v amdgpu > $O/stamps_cfg2.txt)
cat $O/compression_ab.txt | cut -c1-150

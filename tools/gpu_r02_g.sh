#!/bin/bash
mkdir -p gpurun_out/r02g; export TMPDIR=/tmp; O=gpurun_out/r02g; R=$GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
for c in cfg2 cfg3 cfg4 cfg5; do
  (timeout 600 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err)
done
(timeout 300 python bench.py --gpus 2 --same-device --steps 20 --warmup 3 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?" >> $O/bench_2rank.err)
# kernel trace of the default bench command
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-search --no-probes > /tmp/kt_bench.json 2>/tmp/kt.err); cd $R
python tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > $O/kernel_stats_cfg2.txt 2>&1; cp /tmp/kt_bench.json $O/kernel_stats_cfg2_bench_line.json
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o r -- python $R/bench.py --config cfg3 --steps 40 --no-cpu-baseline --no-search --no-probes > /tmp/kt3_bench.json 2>/tmp/kt3.err); cd $R
python tools/rocpd_stats.py $(find /tmp/kt3 -name "*.db" | head -1) > $O/kernel_stats_cfg3.txt 2>&1; cp /tmp/kt3_bench.json $O/kernel_stats_cfg3_bench_line.json
# SQ counters of K2 at cfg2 (3 passes of <= 4 counters)
i=0; for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); cd /tmp && (timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq$i -o r -- python $R/tools/ab_one.py cfg2:10000 cfg3:100000 > /dev/null 2>&1); cd $R
  python tools/rocpd_pmc.py $(find /tmp/sq$i -name "*.db" | head -1) 2>/dev/null | grep -E "k2_prune_mfma4<5, 3>|k2_prune_mfma<1, 4>" >> $O/sq_counters_k2.txt
done
tail -n 6 $O/pytest_gpu.log | cut -c1-200
for c in cfg2 cfg3 cfg4 cfg5 2rank; do echo "== $c"; head -c 300 $O/bench_$c.json; echo; tail -n 2 $O/bench_$c.err | cut -c1-200; done
head -12 $O/kernel_stats_cfg2.txt | cut -c1-200; cat $O/sq_counters_k2.txt | cut -c1-200

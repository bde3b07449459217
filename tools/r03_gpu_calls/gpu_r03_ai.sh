#!/bin/bash
# round 3, call ai: SQ / TCC counters of the final kernels (walk at cfg2 and cfg3, k2c_nodes, the Monte-Carlo-null launch)
mkdir -p gpurun_out/r03ai; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ai; R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/pm$i && (timeout 900 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm$i -o r -- python $R/tools/ab_one.py cfg2:10000 cfg3:100000 > $O/ab_$i.log 2>&1); cd $R
  python tools/rocpd_pmc.py $(find /tmp/pm$i -name "*.db" | head -1) | grep "k2_prune\|k2c_nodes\|k1_build\|k3_score" > $O/pmc_walks_$i.txt 2>&1
  cd /tmp && rm -rf /tmp/pn$i && (timeout 900 rocprofv3 --kernel-trace --pmc $set -d /tmp/pn$i -o r -- python $R/tools/mcnull_one.py 4 > $O/null_$i.log 2>&1); cd $R
  python tools/rocpd_pmc_top.py $(find /tmp/pn$i -name "*.db" | head -1) k2_prune 4 > $O/pmc_null_$i.txt 2>&1
done
head -50 $O/pmc_walks_1.txt | cut -c1-170; head -12 $O/pmc_null_1.txt | cut -c1-170

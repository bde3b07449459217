#!/bin/bash
mkdir -p gpurun_out/r02d; export TMPDIR=/tmp; O=gpurun_out/r02d
export CAFEHIP_K2CFG4=5,3,2,4
(timeout 600 python tools/ab_variants.py main ablB ablA ablAB -- cfg2:10000 > $O/ab_pinned.log 2>&1)
for v in ablAB ablB; do
(CAFEHIP_LIB=tools/_variants/stamps_$v/libcafehip.so timeout 300 python tools/k2_stamps.py cfg2 > $O/stamps_$v.log 2>&1)
done
unset CAFEHIP_K2CFG4
(rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_|SQ_INST_LEVEL|SQ_WAIT_INST|SQ_IFETCH|SQ_BUSY|SQ_WAVE_" | head -80 > $O/counters.txt)
cd /tmp && (timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d /tmp/ic -o r -- python $GRAFT_REPO_ROOT/tools/ab_one.py cfg2:10000 > /dev/null 2>&1); cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/ic -name "*.db" | head -1) 2>/dev/null | grep k2_prune | head -12 > $O/icache.txt
for f in ab_pinned stamps_ablAB stamps_ablB; do echo "=== $f"; grep -v amdgpu $O/$f.log | cut -c1-200; done; echo ==; cat $O/counters.txt | head -60; cat $O/icache.txt

#!/bin/bash
# round 3, call c: k2c_nodes operand prefetch depth sweep
mkdir -p gpurun_out/r03c; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03c
for pf in 0 -1 8 12 16 24 40; do
  echo "== k2c_prefetch=$pf"
  CAFEHIP_K2C_PREFETCH=$pf timeout 600 python tools/ab_one.py cfg2:10000 cfg3:100000 cfg4:62500 2>&1 | grep "^cfg" | cut -c1-420
done > $O/k2c_prefetch_sweep.txt 2>&1
cat $O/k2c_prefetch_sweep.txt

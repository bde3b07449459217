#!/bin/bash
# round 3, call i: K1 columns-per-thread sweep (bit-identical by construction: every entry keeps its own term order)
mkdir -p gpurun_out/r03i; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03i
for q in 8 4 2; do
  echo "== K1Q=$q"
  touch cafe_amd/csrc/k1_matrices.hip
  CAFEHIP_EXTRA_CFLAGS="-DCAFEHIP_K1Q=$q" python -c "from cafe_amd import build; build.build()" 2>&1 | grep -E "error"
  timeout 600 python tools/ab_one.py cfg2:10000 cfg3:100000 cfg4:62500 2>&1 | grep "^cfg" | cut -c1-150
done > $O/k1q_sweep.txt 2>&1
cat $O/k1q_sweep.txt

#!/bin/bash
# round 3, call j: K1 staging merged + K3 one-word publish (bench A/B vs HEAD is implicit: 0.1235-0.127 ms before); k2c ring depth sweep
mkdir -p gpurun_out/r03j; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03j
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compression.py tests/test_gpu_multi_eval.py -q -x 2>&1 | tail -3)
for d in 6 4 8 12 16; do
  echo "== CAFE_K2C_DEPTH=$d"
  touch cafe_amd/csrc/k2c_tables.hip
  CAFEHIP_EXTRA_CFLAGS="-DCAFE_K2C_DEPTH=$d" python -c "from cafe_amd import build; build.build()" 2>&1 | grep -E "error"
  timeout 600 python tools/ab_one.py cfg2:10000 cfg3:100000 cfg4:62500 2>&1 | grep "^cfg" | cut -c1-150
done > $O/k2c_depth_sweep.txt 2>&1
cat $O/k2c_depth_sweep.txt

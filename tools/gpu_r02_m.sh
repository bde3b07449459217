#!/bin/bash
mkdir -p gpurun_out/r02m; export TMPDIR=/tmp; O=gpurun_out/r02m
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x > $O/pytest.log 2>&1); tail -3 $O/pytest.log
(CAFEHIP_K2CFG4=5,3,2,4 timeout 900 python tools/ab_variants.py main d3 d2 d4 main d3 d2 d4 -- cfg2:10000 > $O/ab_cfg2.log 2>&1)
(timeout 1500 python tools/ab_variants.py main d3 d2 d4 -- cfg2:10000 cfg3:100000 cfg4:62500 cfg5:100000 > $O/ab_big.log 2>&1)
grep -v amdgpu $O/ab_cfg2.log | cut -c1-110; grep -v amdgpu $O/ab_big.log | cut -c1-200

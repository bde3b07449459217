#!/bin/bash
mkdir -p gpurun_out/r02p; export TMPDIR=/tmp; O=gpurun_out/r02p
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_multi_eval.py -q -x > $O/pytest.log 2>&1); tail -3 $O/pytest.log
(CAFEHIP_K2CFG4=5,3,2,4 timeout 900 python tools/ab_variants.py main nopf main nopf -- cfg2:10000 > $O/ab_cfg2.log 2>&1)
(timeout 1500 python tools/ab_variants.py main nopf -- cfg3:100000 cfg4:62500 cfg5:100000 > $O/ab_big.log 2>&1)
(CAFEHIP_K2CFG4=5,3,2,4 CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg2 2>&1 | grep -v amdgpu > $O/stamps_cfg2.log)
grep -v amdgpu $O/ab_cfg2.log | cut -c1-110; grep -v amdgpu $O/ab_big.log | cut -c1-200; tail -24 $O/stamps_cfg2.log

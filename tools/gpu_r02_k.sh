#!/bin/bash
mkdir -p gpurun_out/r02k; export TMPDIR=/tmp; O=gpurun_out/r02k
(CAFEHIP_K2CFG4=5,3,2,4 timeout 600 python tools/ab_variants.py main d3 main d3 -- cfg2:10000 > $O/ab_cfg2.log 2>&1)
(timeout 900 python tools/ab_variants.py main d3 -- cfg2:10000 cfg2:3000 cfg2:30000 cfg3:20000 cfg4:10000 > $O/ab_auto.log 2>&1)
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg3 > $O/stamps_cfg3.log 2>&1)
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg4 > $O/stamps_cfg4.log 2>&1)
grep -v amdgpu $O/ab_cfg2.log | cut -c1-110; grep -v amdgpu $O/ab_auto.log | cut -c1-200

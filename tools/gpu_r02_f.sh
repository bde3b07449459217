#!/bin/bash
mkdir -p gpurun_out/r02f; export TMPDIR=/tmp; O=gpurun_out/r02f
export CAFEHIP_K2CFG4=5,3,2,4
(timeout 900 python tools/ab_variants.py main oldgather nocherry noprio minimal minimal_d0 minimal_d4 main -- cfg2:10000 > $O/ab_cfg2.log 2>&1)
unset CAFEHIP_K2CFG4
export CAFEHIP_K2CFG=1,4,1,4
(timeout 900 python tools/ab_variants.py main oldgather nocherry minimal minimal_d0 -- cfg3:100000 > $O/ab_cfg3.log 2>&1)
grep -v amdgpu $O/ab_cfg2.log | cut -c1-120; grep -v amdgpu $O/ab_cfg3.log | cut -c1-120

#!/bin/bash
# round 2, GPU call C: operand-stream ablation, park slots, epilogue with the prior in registers
mkdir -p gpurun_out/r02c; export TMPDIR=/tmp; O=gpurun_out/r02c
(timeout 900 python tools/ab_variants.py main ablB ablA ablAB -- cfg2:10000 cfg3:100000 cfg4:62500 > $O/ab.log 2>&1; echo "rc=$?" >> $O/ab.log)
(CAFEHIP_K2SLOTS=0 timeout 300 python tools/ab_one.py cfg3:100000 cfg4:62500 > $O/noslots.log 2>&1)
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg2 > $O/stamps_cfg2.log 2>&1; echo "rc=$?" >> $O/stamps_cfg2.log)
(timeout 1500 python tools/collect_pmc.py $O/pmc > $O/pmc.log 2>&1; echo "rc=$?" >> $O/pmc.log)
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
for f in ab noslots stamps_cfg2 pmc; do echo "=== $f"; cat $O/$f.log | grep -v amdgpu.ids | cut -c1-230; done
echo "=== pytest"; tail -n 12 $O/pytest_gpu.log

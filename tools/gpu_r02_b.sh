#!/bin/bash
# round 2, GPU call B: restructured K2 (cherry fill, filtered epilogue, unconditional gathers, priorities)
mkdir -p gpurun_out/r02b; export TMPDIR=/tmp; O=gpurun_out/r02b
(timeout 900 python tools/ab_variants.py main noprio d2 d4 d0 -- cfg2:10000 cfg3:100000 cfg4:62500 cfg5:100000 cfg2:1000 cfg2:30000 > $O/ab.log 2>&1; echo "rc=$?" >> $O/ab.log)
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg2 > $O/stamps_cfg2.log 2>&1; echo "rc=$?" >> $O/stamps_cfg2.log)
(timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
for c in cfg2 cfg4 cfg5; do
  (timeout 600 python bench.py --config $c --steps 40 --warmup 5 > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err)
done
for f in ab stamps_cfg2; do echo "=== $f"; cat $O/$f.log | grep -v amdgpu.ids; done
echo "=== pytest"; tail -n 30 $O/pytest_gpu.log
for c in cfg2 cfg4 cfg5; do echo "=== bench $c"; head -c 600 $O/bench_$c.json; echo; tail -n 3 $O/bench_$c.err; done

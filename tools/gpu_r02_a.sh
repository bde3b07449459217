#!/bin/bash
# round 2, GPU call A: A/B of the k-loop pipelining variants, phase timeline, bench lines of all configs, GPU tests
mkdir -p gpurun_out/r02a; export TMPDIR=/tmp; O=gpurun_out/r02a
(timeout 1200 python tools/ab_variants.py main d0 d2 d4 d3ni -- cfg2:10000 cfg3:100000 cfg4:62500 cfg5:100000 > $O/ab.log 2>&1; echo "rc=$?" >> $O/ab.log)
for v in stamps stamps0; do
  (CAFEHIP_LIB=tools/_variants/$v/libcafehip.so timeout 300 python tools/k2_stamps.py cfg2 > $O/stamps_cfg2_$v.log 2>&1; echo "rc=$?" >> $O/stamps_cfg2_$v.log)
done
(CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 300 python tools/k2_stamps.py cfg3 > $O/stamps_cfg3.log 2>&1; echo "rc=$?" >> $O/stamps_cfg3.log)
for c in cfg2 cfg3 cfg4 cfg5; do
  (timeout 600 python bench.py --config $c --steps 40 --warmup 5 > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err)
done
(timeout 300 python bench.py --gpus 2 --same-device --steps 20 --warmup 3 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?" >> $O/bench_2rank.err)
(timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
for f in ab stamps_cfg2_stamps stamps_cfg2_stamps0 pytest_gpu; do echo "=== $f"; tail -n 25 $O/$f.log; done
for c in cfg2 cfg3 cfg4 cfg5 2rank; do echo "=== bench $c"; tail -c 1500 $O/bench_$c.json; tail -n 3 $O/bench_$c.err; done

#!/bin/bash
mkdir -p gpurun_out/r02e; export TMPDIR=/tmp; O=gpurun_out/r02e
(timeout 900 python -m pytest tests/test_gpu_multi_eval.py tests/test_gpu_native_comm.py -x -q -s > $O/pytest_multi.log 2>&1; echo "rc=$?" >> $O/pytest_multi.log)
(timeout 600 python tools/speculation_timing.py > $O/speculation.log 2>&1; echo "rc=$?" >> $O/speculation.log)
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
for c in cfg2 cfg3 cfg4 cfg5; do
  (timeout 600 python bench.py --config $c --steps 40 --warmup 5 > $O/bench_$c.json 2> $O/bench_$c.err; echo "rc=$?" >> $O/bench_$c.err)
done
echo "=== multi"; tail -n 25 $O/pytest_multi.log | cut -c1-250; echo "=== spec"; grep -v amdgpu $O/speculation.log | cut -c1-250; echo "=== all"; tail -n 8 $O/pytest_gpu.log | cut -c1-200
for c in cfg2 cfg3 cfg4 cfg5; do echo "== $c"; head -c 400 $O/bench_$c.json; echo; tail -n 2 $O/bench_$c.err; done

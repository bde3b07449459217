#!/bin/bash
# one GPU round: smoke, GPU tests, bench (logs under gpurun_out/)
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
(timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(timeout 600 python bench.py --steps ${STEPS:-20} --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log)
for f in smoke pytest_gpu bench; do echo "=== $f"; tail -n 12 gpurun_out/$f.log; done

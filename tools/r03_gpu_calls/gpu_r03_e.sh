#!/bin/bash
# round 3, call e: after the comm layer + first-evaluation ordering fixes: whole GPU suite + bench
mkdir -p gpurun_out/r03e; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03e
(timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -25 $O/pytest_gpu.log | cut -c1-220
(timeout 600 python bench.py --no-cpu-baseline --no-search --no-probes > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?" >> $O/bench_cfg2.err); head -c 300 $O/bench_cfg2.json; echo

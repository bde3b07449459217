#!/bin/bash
mkdir -p gpurun_out/r02h; export TMPDIR=/tmp; O=gpurun_out/r02h
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log)
(timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
(timeout 300 python bench.py --gpus 1 --force-dist --steps 50 --warmup 5 --no-cpu-baseline --no-search --no-probes > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "rc=$?" >> $O/bench_forcedist.err)
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
tail -n 3 $O/smoke.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -B5 "Error\|FAILED" $O/pytest_gpu.log | head -40
head -c 700 $O/bench_forcedist.json; echo; tail -n 3 $O/bench_forcedist.err; head -c 400 $O/bench.json; echo

#!/bin/bash
# round 3, call f: the rewritten bench.py: default N=1 run, forced one-rank native exchange, 2 ranks on one device
mkdir -p gpurun_out/r03f; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03f
(time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err) 2>> $O/bench_default.err; tail -5 $O/bench_default.err | cut -c1-300; head -c 700 $O/bench_default.json; echo
(timeout 600 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-search --no-probes --no-strong > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "rc=$?" >> $O/bench_forcedist.err); tail -3 $O/bench_forcedist.err | cut -c1-300; python -c "
import json; d=json.load(open('$O/bench_forcedist.json')); print({k:d[k] for k in ('value','ms_per_step','rccl_ranks','comm')}, d.get('exchange'))"
(timeout 900 python bench.py --gpus 2 --same-device --no-cpu-baseline --no-search --no-probes > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?" >> $O/bench_2rank.err); tail -3 $O/bench_2rank.err | cut -c1-300; python -c "
import json; d=json.load(open('$O/bench_2rank.json')); print({k:d[k] for k in ('value','ms_per_step','rccl_ranks','comm')}, d.get('exchange'), d.get('strong_scaling'))"

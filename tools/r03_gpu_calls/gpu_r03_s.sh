#!/bin/bash
# round 3, call s: trimmed products with operand rings as deep as the registers of the full product allow
mkdir -p gpurun_out/r03s; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03s
(timeout 900 python -m pytest tests/test_gpu_batch_trim.py tests/test_gpu_full_size.py -m gpu -q -k "trim or null or batch" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log | cut -c1-200
(timeout 600 python tools/mcnull_one.py 6 16:1,4,1,4 > $O/mcnull.log 2>&1); grep "^mcnull" $O/mcnull.log | cut -c1-330
(timeout 600 python bench.py --config cfg3 --steps 60 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables > $O/bench_cfg3.json 2> $O/bench_cfg3.err); head -c 400 $O/bench_cfg3.json; echo

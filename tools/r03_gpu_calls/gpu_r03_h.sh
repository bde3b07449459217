#!/bin/bash
# round 3, call h: new full-size tests + bench contract
mkdir -p gpurun_out/r03h; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03h
(timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_bench_contract.py -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -25 $O/pytest.log | cut -c1-250

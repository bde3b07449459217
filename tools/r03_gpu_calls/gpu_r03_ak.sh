#!/bin/bash
# round 3, call ak: trimmed park fetch in batch mode: tests, time and fabric traffic of the Monte-Carlo-null launch
mkdir -p gpurun_out/r03ak; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ak
(timeout 1500 python -m pytest tests/test_gpu_batch_trim.py tests/test_gpu_full_size.py tests/test_gpu_host_driver.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log | cut -c1-300
(timeout 600 python tools/mcnull_one.py 6 > $O/mcnull.log 2>&1); grep "^mcnull" $O/mcnull.log | cut -c1-120
(timeout 900 python tools/pmc_mcnull.py trimmed-park-fetch > $O/pmc.log 2>&1); tail -1 $O/pmc.log

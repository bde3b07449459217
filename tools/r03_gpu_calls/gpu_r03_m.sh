#!/bin/bash
# round 3, call m: batch launches with one product instantiation per trimmed row-tile count
mkdir -p gpurun_out/r03m; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03m; R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_batch_trim.py tests/test_gpu_full_size.py -m gpu -q -k "trim or null or batch" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -6 $O/pytest.log | cut -c1-200
(timeout 600 python tools/mcnull_one.py 6 > $O/mcnull.log 2>&1); tail -2 $O/mcnull.log
(CAFEHIP_BATCH_LOCKSTEP=0 timeout 600 python tools/mcnull_one.py 6 > $O/mcnull_nolock.log 2>&1); tail -1 $O/mcnull_nolock.log
(CAFEHIP_BATCH_TRIM=0 timeout 600 python tools/mcnull_one.py 6 > $O/mcnull_notrim.log 2>&1); tail -1 $O/mcnull_notrim.log

#!/bin/bash
# round 3, call d: native communicator (direct exchange / RCCL) on one GPU
mkdir -p gpurun_out/r03d; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03d
(timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_native_comm.py tests/test_gpu_multi_rank.py -q -x > $O/pytest_comm.log 2>&1; echo "rc=$?" >> $O/pytest_comm.log); tail -40 $O/pytest_comm.log

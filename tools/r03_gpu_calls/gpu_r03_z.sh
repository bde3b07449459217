#!/bin/bash
# round 3, call z: the configs[4] pipeline through the host driver (lambda -s + report), phase times; report tests after the bulk random draw
mkdir -p gpurun_out/r03z; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03z
(timeout 1500 python -m pytest tests/test_gpu_host_driver.py tests/test_gpu_full_size.py tests/test_gpu_transcript3.py tests/test_gpu_multi_rank.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -4 $O/pytest.log | cut -c1-300
(CAFEHOST_TIMING=1 timeout 900 python tools/cfg5_pipeline_time.py 100000 > $O/pipeline.log 2>&1); grep -v amdgpu.ids $O/pipeline.log | cut -c1-200
(CAFEHOST_TIMING=1 timeout 900 python tools/cfg5_pipeline_time.py 100000 > $O/pipeline2.log 2>&1); grep "null:\|report" $O/pipeline2.log | cut -c1-200

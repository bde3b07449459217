#!/bin/bash
# round 3, call g: batch-mode trimming: parity + the cfg5 Monte-Carlo-null launch
mkdir -p gpurun_out/r03g; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03g
(timeout 1200 python -m pytest tests/test_gpu_batch_trim.py tests/test_gpu_parity.py "tests/test_gpu_full_size.py::test_cfg5_null_sharded_by_root_size_recombines_bit_exactly" -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -15 $O/pytest.log | cut -c1-250
for tr in 1 0; do CAFEHIP_BATCH_TRIM=$tr timeout 600 python tools/mcnull_one.py 6 2>&1 | grep mcnull | cut -c1-330; done | tee $O/mcnull.txt

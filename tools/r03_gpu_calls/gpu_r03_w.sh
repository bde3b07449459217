#!/bin/bash
# round 3, call w: tests added since call v; cfg5 and cfg2 bench lines re-taken so that roofline.traffic reads this round's PMC file
mkdir -p gpurun_out/r03w; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03w
(timeout 1500 python -m pytest tests/test_gpu_batch_trim.py tests/test_gpu_host_driver.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -5 $O/pytest.log | cut -c1-300
(timeout 900 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "rc=$?" >> $O/bench_cfg5.err); head -c 200 $O/bench_cfg5.json; echo
(timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?" >> $O/bench_cfg2.err); head -c 200 $O/bench_cfg2.json; echo

#!/bin/bash
# round 3, call aj: steps with two gathered children request both at once (4x4x4 walk), against the previous build, same box
mkdir -p gpurun_out/r03aj; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03aj
(timeout 1500 python -m pytest tests/test_gpu_compression.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_multi_eval.py tests/test_gpu_batch_trim.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log | cut -c1-300
for rep in 1 2 3 4; do for v in new old; do
  if [ $v = new ]; then unset CAFEHIP_LIB; else export CAFEHIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/oldll/libcafehip.so; fi
  python bench.py --steps 300 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables 2>/dev/null | python -c "
import json,sys,re; d=json.load(sys.stdin); r=d['roofline']; m=re.search(r'cfg\(nftw,nrtw,wf,wr\)=\S+', d['engine']); print('$v cfg2', round(d['ms_per_step'],4), round(r['avg_launch_ms'],4), m.group(0))"
done; done
for v in new old; do
  if [ $v = new ]; then unset CAFEHIP_LIB; else export CAFEHIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/oldll/libcafehip.so; fi
  python bench.py --table test1 --steps 300 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables 2>/dev/null | python -c "
import json,sys,re; d=json.load(sys.stdin); r=d['roofline']; print('$v test1', round(d['ms_per_step'],4), round(r['avg_launch_ms'],4))"
done

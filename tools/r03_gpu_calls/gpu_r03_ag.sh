#!/bin/bash
# round 3, call ag: the walk's set-up loads requested together (k2_fill_setup) against the previous build, same box
mkdir -p gpurun_out/r03ag; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03ag
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log | cut -c1-300
for rep in 1 2 3; do for v in new old; do
  if [ $v = new ]; then unset CAFEHIP_LIB; else export CAFEHIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/oldpro/libcafehip.so; fi
  for c in cfg2 cfg4; do
  python bench.py --config $c --steps 300 --no-cpu-baseline --no-search --no-probes --no-strong --no-tables 2>/dev/null | python -c "
import json,sys,re; d=json.load(sys.stdin); r=d['roofline']; m=re.search(r'cfg\(nftw,nrtw,wf,wr\)=\S+', d['engine']); print('$v $c', round(d['ms_per_step'],4), round(r['avg_launch_ms'],4), m.group(0))"
  done
done; done

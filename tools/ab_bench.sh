#!/bin/bash
# A/B of two builds on one box: tools/ab_bench.sh <libA> <libB> [configs...]  -- per config: ms per evaluation, walk launch ms,
# table launches ms, Monte-Carlo-null launch ms (cfg5); two passes each, interleaved
export TMPDIR=/tmp
A=$1; B=$2; shift 2
CFGS=${@:-cfg2 cfg3 cfg4 cfg5}
for pass in 1 2; do for c in $CFGS; do for lib in $A $B; do
  CAFEHIP_LIB=$lib timeout 600 python bench.py --config $c --steps 100 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | \
  python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s %s pass $pass: step %.4f ms  walk %.4f (frac %.3f)  tables %.4f  null %s  engine cfg %s' % ('$lib'[-40:], '$c', d['ms_per_step'], r['avg_launch_ms'], r['frac'], (r['factor_tables'] or {}).get('ms_per_evaluation',0), ('%.3f' % d['mc_null']['launch_ms']) if 'mc_null' in d else '-', d['engine'].split('cfg(nftw,nrtw,wf,wr)=')[1].split()[0]))"
done; done; done

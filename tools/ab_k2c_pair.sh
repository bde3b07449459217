mkdir -p gpurun_out/r05
export TMPDIR=/tmp
CAFEHIP_K2C_PAIR=1 timeout 900 python -m pytest tests/test_gpu_compression.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for pass in 1 2; do for c in cfg3 cfg4 cfg2 cfg5; do for pr in 0 1; do
  CAFEHIP_K2C_PAIR=$pr timeout 600 python bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | \
  python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pair=$pr %s pass $pass: step %.4f ms  walk %.4f (frac %.3f)  tables %.4f (frac %.3f)  null %s  score %r' % ('$c', d['ms_per_step'], r['avg_launch_ms'], r['frac'], (r['factor_tables'] or {}).get('ms_per_evaluation',0), (r['factor_tables'] or {}).get('frac',0), ('%.3f' % d['mc_null']['launch_ms']) if 'mc_null' in d else '-', d.get('last_score')))"
done; done; done

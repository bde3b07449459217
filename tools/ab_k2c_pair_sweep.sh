# where do paired row tiles start to pay?  configs[1]'s shape at growing table sizes, option k2c_pair forced off / on
export TMPDIR=/tmp
for F in 20000 40000 80000 160000; do for pr in 0 1 0 1; do
  CAFEHIP_K2C_PAIR=$pr timeout 600 python bench.py --config cfg2 --families $F --steps 60 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | \
  python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg2 F=$F pair=$pr: step %.4f ms  walk %.4f  tables %.4f (frac %.3f)  %s' % (d['ms_per_step'], r['avg_launch_ms'], (r['factor_tables'] or {}).get('ms_per_evaluation',0), (r['factor_tables'] or {}).get('frac',0), d['engine'].split('level_tiles=')[1]))"
done; done

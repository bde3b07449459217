export TMPDIR=/tmp
run() { CAFEHIP_COMPRESS_THETA=$2 timeout 600 python bench.py --config $1 $3 --steps 40 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; ft=r.get('factor_tables') or {}
print('$1 $3 theta=$2: step %.4f ms  walk %.4f  tables %.4f  %s' % (d['ms_per_step'], r['avg_launch_ms'], ft.get('ms_per_evaluation',0), d['engine'].split('compressed(')[1] if 'compressed(' in d['engine'] else '-'))"; }
for p in 1 2; do for th in 0.5 0.7 0.8 0.9 1.0; do run cfg4 $th; done; done
for th in 0.5 0.7 0.8 0.9; do run cfg2 $th "--families 160000"; done
for th in 0.5 0.7 0.8 0.9; do run cfg2 $th "--families 40000"; done

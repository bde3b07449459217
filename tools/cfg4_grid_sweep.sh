export TMPDIR=/tmp
CAFEHIP_K2TUNE_LOG=1 timeout 300 python bench.py --config cfg4 --steps 40 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>&1 >/dev/null | grep "wave grid"
for g in 1,5,2,2 1,3,2,4 1,4,1,3 1,4,2,3 1,3,1,4 1,5,1,2 2,3,1,4 1,2,2,5 1,2,1,5 2,5,1,2 1,4,4,3; do
CAFEHIP_K2CFG=$g CAFEHIP_MFMA=16 timeout 300 python bench.py --config cfg4 --steps 40 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('k2cfg=$g: step %.4f walk %.4f frac %.3f  %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['engine'].split('k2:')[1].split(' park')[0]))
except Exception as e: print('k2cfg=$g: failed', e)"
done

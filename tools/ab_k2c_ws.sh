# producer / consumer waves in the table kernel (k2c_nodes_ws, option k2c_ws = consecutive tiles per workgroup) against the
# paired kernel: parity tests with it forced on every level, then bench.py per configuration
export TMPDIR=/tmp
L=${1:-tools/_variants/ws/libcafehip.so}
CAFEHIP_LIB=$L CAFEHIP_K2C_WS=3 CAFEHIP_K2C_PAIR=1 timeout 900 python -m pytest tests/test_gpu_compression.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for pass in 1 2; do for c in cfg3 cfg4 cfg5; do for ws in 0 2 4 8 16; do
  CAFEHIP_LIB=$L CAFEHIP_K2C_WS=$ws timeout 600 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; ft=r.get('factor_tables') or {}
print('ws=%-2s $c pass $pass: step %.4f ms  walk %.4f  tables %.4f (frac %.3f)' % ('$ws', d['ms_per_step'], r['avg_launch_ms'], ft.get('ms_per_evaluation',0), ft.get('frac',0)))"
done; done; done

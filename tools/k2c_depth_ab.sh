mkdir -p gpurun_out/r05; export TMPDIR=/tmp
for c in cfg3 cfg4; do for lib in cafe_amd/lib tools/_variants/k2cd2 tools/_variants/k2cd3 tools/_variants/k2cd6 cafe_amd/lib; do
  CAFEHIP_LIB=$lib/libcafehip.so timeout 600 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; ft=r.get('factor_tables') or {}
print('%-28s $c: step %.4f ms  walk %.4f  tables %.4f (frac %.3f)' % ('$lib', d['ms_per_step'], r['avg_launch_ms'], ft.get('ms_per_evaluation',0), ft.get('frac',0)))"
done; done
K2C_STAMPS_DETAIL=1 CAFEHIP_LIB=tools/_variants/stamps/libcafehip.so timeout 250 python tools/k2c_stamps.py cfg3 2>&1 | tail -18

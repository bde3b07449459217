mkdir -p gpurun_out/r06
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
tail -3 gpurun_out/r06/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/bench_default.json'))
print({k:d.get(k) for k in ('value','ms_per_step','bench_wall_s')})
print(d.get('ms_per_step_blocks'))
r=d['roofline']; print('walk', r['avg_launch_ms'], r['frac'], 'tables', r['factor_tables']['ms_per_evaluation'], r['factor_tables']['frac'], 'whole', r['whole_evaluation'])
for k,v in d.get('configs',{}).items():
    if isinstance(v,dict) and 'roofline' in v: print('cfg',k, v['ms_per_step'], v['value'], v['roofline']['frac'], v['roofline']['factor_tables']['frac'], v['roofline']['whole_evaluation']['frac'], v.get('table_generation_s'))
s=d.get('strong_scaling'); print('strong', s['ms_per_step'], s['value'], (s.get('roofline') or {}).get('frac'))
for k,v in d.get('tables',{}).items():
    if isinstance(v,dict) and 'ms_per_step' in v: print('table',k,v['ms_per_step'], v['roofline']['whole_evaluation']['frac'])
ls=d.get('lambda_search',{}); print('search', {k: ls.get(k) for k in ('wall_s','search_s','search_s_without_lookahead','cold_process_wall_s','cold_process_search_s','same_result_in_cold_process','evaluations')})
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
timeout 1500 python tools/strong_blocks.py > gpurun_out/r06/strong_blocks.txt 2>&1
tail -50 gpurun_out/r06/strong_blocks.txt

mkdir -p gpurun_out/final3
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/final3/pytest_gpu.txt 2>&1; tail -3 gpurun_out/final3/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/final3/bench_cfg2.json 2> gpurun_out/final3/bench_cfg2.err; tail -c 300 gpurun_out/final3/bench_cfg2.err
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --steps 60 --warmup 5 --no-search --no-tables --no-strong > gpurun_out/final3/bench_$c.json 2>/dev/null; done
python - <<'PY'
import json
for c in ("cfg2","cfg3","cfg4","cfg5"):
    d=json.load(open("gpurun_out/final3/bench_%s.json"%c)); r=d["roofline"]; ft=r.get("factor_tables") or {}
    print(c, "step %.4f ms value %.2f M/s walk %.4f frac %.3f tables %.4f frac %.3f whole %.3f" % (d["ms_per_step"], d["value"]/1e6, r["avg_launch_ms"], r["frac"], ft.get("ms_per_evaluation",0), ft.get("frac",0), r["whole_evaluation"]["frac"]), d.get("mc_null",{}).get("launch_ms"))
PY

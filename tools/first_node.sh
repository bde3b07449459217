#!/bin/bash
# first_node.sh -- the first thing to run when a node with more than one GPU appears (round 5).
# Nothing in section 4 of DESIGN.md has run between two physical GPUs yet; this script runs, in order and with bounded
# time-outs, exactly what the driver's scaling run will need:
#   1. the rank-per-device parity test (rank r on device r, direct exchange and RCCL): bit-identical to one context
#   2. bench.py --gpus N for N = 2 .. number of GPUs: the weak headline + the strong leg; exchange microseconds in both modes,
#      per-rank step times, skew, and last_score of the strong leg (the same bits for every N)
#   3. the host driver's own launcher (cafehip --gpus N) on the reference's test1 table: the transcript must equal the
#      one-GPU transcript
# On a one-GPU box: first_node.sh --same-device runs the same with every rank on device 0 (a functional dry run: the
# numbers then say nothing about xGMI; N defaults to 8).
# Output: gpurun_out/first_node/ (logs + one JSON per N) and a summary on stdout.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
SAME=""
N_MAX=""
for a in "$@"; do
  case "$a" in
    --same-device) SAME="--same-device";;
    [0-9]*) N_MAX="$a";;
  esac
done
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
if [ -z "$N_MAX" ]; then
  if [ -n "$SAME" ]; then N_MAX=8; else N_MAX=$NGPU; fi
fi
OUT=gpurun_out/first_node
mkdir -p $OUT
echo "devices visible: $NGPU; ranks up to $N_MAX ${SAME:+(all on device 0)}"
if [ -z "$SAME" ] && [ "$NGPU" -lt 2 ]; then
  echo "one GPU only: use --same-device for the functional dry run"; exit 2
fi

echo "== 1. rank-per-device parity test (skips on one GPU)"
timeout 900 python -m pytest tests/test_gpu_comm.py -q -x -k "distinct_devices or one_device" 2>&1 | tail -3

echo "== 1b. RCCL first (the mode north_star names, and the one the default direct exchange leaves least exercised): N=2 with --comm-mode rccl"
if [ -z "$SAME" ] && [ "$N_MAX" -ge 2 ]; then
  timeout 900 python bench.py --gpus 2 --comm-mode rccl --steps 20 --warmup 5 --no-cpu-baseline --no-tables --no-search --no-probes --no-strong \
      > $OUT/bench_2_rccl.json 2> $OUT/bench_2_rccl.err
  echo "   rc=$?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_2_rccl.json')); x=d.get('exchange') or {}; print(d.get('error') or ('%.4f ms/step, mode %s, rccl_ranks %s' % (d['ms_per_step'], x.get('mode'), d.get('rccl_ranks'))))" 2>&1 | tail -1)"
else
  echo "   skipped (RCCL refuses two ranks on one device)"
fi

echo "== 2. bench.py --gpus N"
for n in 1 2 4 8; do
  [ "$n" -gt "$N_MAX" ] && continue
  timeout 1200 python bench.py --gpus $n $SAME --steps 20 --warmup 5 --no-cpu-baseline --no-tables --no-search --no-probes \
      > $OUT/bench_${n}.json 2> $OUT/bench_${n}.err
  rc=$?
  python - $OUT/bench_${n}.json $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("   N=? rc=%s: no JSON line (%s)" % (sys.argv[2], e)); sys.exit(0)
if "error" in d:
    print("   N=%s rc=%s ERROR: %s" % (d.get("n_gpus"), sys.argv[2], d["error"])); sys.exit(0)
x = d.get("exchange") or {}
s = d.get("strong_scaling") or {}
print("   N=%d rc=%s: %.4f ms/step (%.1f M family-evals/s); announced %.4f; mode %s, exchange %.1f us%s; ranks %s, skew %.4f ms;"
      " strong leg %.3f ms, last_score %s"
      % (d["n_gpus"], sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("speculated_hit_ms_per_step", float("nan")), x.get("mode"),
         1e3 * x.get("exchange_ms_per_step", float("nan")),
         (" (RCCL %.1f us)" % (1e3 * x["rccl"]["exchange_ms_per_step"])) if isinstance(x.get("rccl"), dict) and "exchange_ms_per_step" in x["rccl"] else "",
         d.get("comm_world", 1), (d.get("rank_skew_ms_per_step") or {}).get("max_minus_min", 0.0), s.get("ms_per_step", float("nan")),
         float(s["last_score"]).hex() if "last_score" in s else None))
PY
done

echo "== 3. cafehip --gpus N on the reference's test1 table"
T=$(mktemp -d)
python - $T <<'PY'
import gzip, json, os, sys
d = sys.argv[1]
root = os.getcwd()
open(os.path.join(d, "test1_families.txt"), "wb").write(gzip.open(os.path.join(root, "tests/golden/test1_families.txt.gz")).read())
tr = json.load(open(os.path.join(root, "tests/golden/transcripts.json")))
open(os.path.join(d, "test1.sh"), "w").write("\n".join([
    "seed 10", "tree " + tr["test1"]["newick"], "load -i %s -max_size 20" % os.path.join(d, "test1_families.txt"), "lambda -s", ""]))
PY
timeout 600 cafe_amd/bin/cafehip $T/test1.sh > $OUT/test1_1gpu.log 2> $OUT/test1_1gpu.err
for n in 2 8; do
  [ "$n" -gt "$N_MAX" ] && continue
  t0=$(date +%s.%N)
  timeout 900 cafe_amd/bin/cafehip --gpus $n $SAME $T/test1.sh > $OUT/test1_${n}gpu.log 2> $OUT/test1_${n}gpu.err
  rc=$?
  python -c "print('%.2f s' % ($(date +%s.%N) - $t0))" > $OUT/test1_${n}.time
  if cmp -s $OUT/test1_1gpu.log $OUT/test1_${n}gpu.log; then same="transcript == the one-GPU transcript"; else same="TRANSCRIPT DIFFERS"; fi
  echo "   N=$n rc=$rc $(cat $OUT/test1_${n}.time): $same; $(grep 'cafehip:' $OUT/test1_${n}gpu.err | tail -1)"
done
grep -h "Lambda Search Result" -A1 $OUT/test1_1gpu.log | tail -2
rm -rf $T

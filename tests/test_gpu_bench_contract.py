"""bench.py's output contract: ONE JSON line on stdout with the keys the driver reads, the roofline object of the
dominant kernel and the CPU baseline timed beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "2"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["unit"] == "family-evals/s" and d["value"] > 1e6          # north-star floor: 10^6 evaluations/s
    assert abs(d["value"] - d["config"]["families_per_gpu"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    # the parity gate recorded with the timing: per-family log posterior vs the oracle
    assert c["gpu_vs_oracle_max_rel_err_log_posterior"] < 1e-6
    # round 3 additions: set-up reported apart, traffic next to what the launch inherently needs, other tables
    assert d["setup_ms"]["set_families_detail_ms"]["total"] > 0
    assert r["traffic_minimal_bytes"] > 0 and "whole_evaluation" in r
    assert d["strong_scaling"]["families_total"] == 8 * 62464 and d["strong_scaling"]["value"] > 1e6
    assert d["tables"]["test1"]["families"] == 14787 and d["tables"]["turnover"]["value"] > 1e6
    # round 4: the headline table with compression off is the floor of "the headline is a property of the table"
    u = d["tables"]["uncompressed"]
    assert u["options"] == {"compress": 0} and "used=1" not in u["engine"]
    assert u["value"] < d["value"]
    assert c["matrix_build_s"] > 0 and c["family_loop_value"] > c["value"]
    # round 6: the headline is the median of five timed regions, printed with its spread; every single-GPU configuration of
    # BASELINE.json is a leg of the same line with its own roofline; the strong leg carries one too; the credit ratios sit beside
    # the fraction and exceed 1 (they are not fractions); the first search of a fresh process is timed beside the warm one
    b = d["ms_per_step_blocks"]
    assert len(b["ms_per_step"]) == 5 and b["min"] <= d["ms_per_step"] <= b["max"]
    for key, fams in (("2", 100000), ("4", 100000)):
        leg = d["configs"][key]
        assert leg["families"] == fams and leg["value"] > 1e6 and 0 < leg["roofline"]["frac"] <= 1
        assert leg["roofline"]["issued_flops_per_launch"] > 0 and leg["roofline"]["avg_launch_ms"] > 0
        assert len(leg["ms_per_step_blocks"]["ms_per_step"]) == 3
    assert d["configs"]["1"]["ms_per_step"] == d["ms_per_step"]
    assert 0 < d["strong_scaling"]["roofline"]["frac"] <= 1
    assert "useful_frac" not in r
    assert d["algorithmic_credit"]["F_alg_over_fp64_peak"] > 1 and d["algorithmic_credit"]["B_alg_over_hbm_peak"] > 1
    ls = d["lambda_search"]
    assert ls["cold_process_search_s"] > 0 and ls["same_result_in_cold_process"] is True
    assert d["bench_wall_s"] < 240


def test_forced_one_rank_run_goes_through_the_native_exchange():
    # exactly what the driver starts for N > 1, with one rank: the timed step is cafehip_eval_posterior_sharded (direct
    # exchange inside the score kernel), RCCL timed beside it; the score must carry the same bits as the plain run
    def run(extra):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "24", "--warmup", "2",
                              "--no-cpu-baseline", "--no-search", "--no-probes", "--no-strong", "--no-tables"] + extra,
                             cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1
        return json.loads(lines[0])
    plain = run([])
    forced = run(["--force-dist", "--comm", "native"])
    assert forced["comm"] == "native" and forced["n_gpus"] == 1 and forced["comm_world"] == 1
    # the timed region ran in direct mode: RCCL was never loaded, and the record says so (rccl_ranks used to be
    # hard-wired to the torch world size); what RCCL itself reports appears with the comparison leg that initialises it
    assert forced["rccl_ranks"] == 0 and forced["rccl_initialised"] is False
    st = forced["comm_status_per_rank"][0]
    assert st["mode_agreed_at_init"] == "direct" and st["peers_heard_by_probe"] == 1
    assert len(forced["per_rank_ms_per_step"]) == 1 and forced["rank_skew_ms_per_step"]["max_minus_min"] == 0
    x = forced["exchange"]
    assert x["mode"] == "direct" and x["exchange_ms_per_step"] < 0.020        # <= 20 us with one rank
    assert "rccl" in x and ("error" in x["rccl"] or (x["rccl"]["exchange_ms_per_step"] > 0 and x["rccl"]["rccl_ranks"] == 1
                                                      and x["rccl"]["rccl_initialised"]))
    assert forced["config"]["last_score"] == plain["config"]["last_score"]     # same step, same table: same bits
    assert forced["ms_per_step"] < 1.25 * plain["ms_per_step"]


def test_native_exchange_failure_falls_back_to_the_torch_path_on_every_rank(monkeypatch):
    # a node where the library's exchange cannot be set up must still produce the line (through torch.distributed),
    # and say so: the failure is injected on rank 0 of a forced one-rank run
    env = dict(os.environ, BENCH_FORCE_NATIVE_FAILURE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "16", "--warmup", "2",
                          "--no-cpu-baseline", "--no-search", "--no-probes", "--no-strong", "--no-tables"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert d["comm"] == "torch" and "forced by BENCH_FORCE_NATIVE_FAILURE" in d["comm_fallback"]
    assert d["value"] > 1e6 and d["exchange"]["mode"].startswith("torch.distributed")


def test_round5_keys_and_the_error_line_of_a_failing_job():
    # the serial headline and, beside it (never folded in), the same steps with the next parameter set announced ahead
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "2", "--no-cpu-baseline",
                          "--no-search", "--no-probes", "--no-strong", "--no-tables"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    sp = d["speculated"]
    assert d["speculated_hit_ms_per_step"] > 0 and sp["last_score_equals_an_evaluation_that_builds_its_matrices"] is True
    assert sp["this_rank"]["hits"] >= 16 and d["ms_per_step"] > 0
    # a rank that fails: rank 0 still prints exactly one line, with an `error` key (two ranks sharing this box's GPU)
    env = dict(os.environ, BENCH_FAIL_RANK="1", CAFEHIP_COMM_TIMEOUT_S="20")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--steps", "8", "--warmup", "2",
                          "--no-cpu-baseline", "--no-search", "--no-probes", "--no-strong", "--no-tables"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    e = json.loads(lines[0])
    assert "error" in e and e["value"] is None and e["n_gpus"] == 2

#!/usr/bin/env python
"""bench.py -- family-likelihood evaluations per second of the HIP hot path.

A "step" is ONE objective evaluation (reset_birthdeath_cache + get_posterior,
cafe/cafe_main.c:319-326 + cafe/lambda.cpp:691-724) over the whole count table of
BASELINE.json configs[1]: 10k synthetic families per GPU, 16-taxon tree, max family size 100,
single lambda.  Every step uses a different lambda (as the optimiser does), builds all
transition matrices, prunes every family, reduces the score and returns it to the host.

    python bench.py --gpus N --steps K --warmup W

For N > 1 run under torch.distributed.run: one process per GPU, families sharded (weak scaling:
10k families per GPU), one RCCL all-gather of the per-chunk partial sums + one all-reduce(min) of
the first-zero index per step.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak


def algorithmic_bytes_per_family(n_leaves, R, C):
    """SURVEY.md section 8(d) B_alg: the reference streams a dense sub-matrix on every child edge:
    8 * (2*R*C + (E-2)*C*C) bytes per family evaluation, E = 2n-2 edges."""
    E = 2 * n_leaves - 2
    return 8.0 * (2.0 * R * C + (E - 2.0) * C * C)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--families", type=int, default=None, help="families per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="skip the lambda-search wall-clock leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for debugging)")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: take the multi-rank code path (process group + packed all_gather) even with 1 rank")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: all ranks share GPU 0 (functional check of the N>1 path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d processes (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    import cafe_amd
    from cafe_amd import distributed as D
    from cafe_amd import prior as cprior
    from cafe_amd import synth
    from cafe_amd import tree as ctree

    # ---- workload: this rank's shard of the family table -------------------------------
    cfg = dict(synth.CONFIGS[args.config])
    F_local = args.families or cfg["F"]
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
    tree = ctree.CafeTree(newick)
    counts = synth.simulate_families(tree, F_local, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1 + rank)
    rng = cafe_amd.init_family_size(cfg["m"])
    R = rng.root_max - rng.root_min + 1
    C = rng.max - rng.min + 1
    lam_p = cprior.poisson_lambda_mle(counts)
    prior = cprior.prior_rfsize_poisson(rng.root_min, lam_p)

    eng = cafe_amd.Engine(local_rank)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    tree.apply(eng)
    eng.set_families(counts, rng)
    n_chunks = eng.num_chunks()
    eng.enable_timing(False)

    packed, p_chunks, p_fz = D.packed_buffer(torch, n_chunks, "cuda")
    gathered = torch.zeros((n_chunks + 1) * world, dtype=torch.float64, device="cuda")
    gathered_host = torch.zeros((n_chunks + 1) * world, dtype=torch.float64).pin_memory()
    bounds = [(r * F_local, (r + 1) * F_local) for r in range(world)]
    has_mu = cfg["mu"] >= 0

    def node_rates(step):
        lam = cfg["lam"] * (1.0 + 0.003 * (step % 97))
        nl = np.full(tree.n_nodes, lam)
        nm = np.full(tree.n_nodes, cfg["mu"] * (1.0 + 0.002 * (step % 89)) if has_mu else -1.0)
        return nl, nm

    kernel_ms = []

    # HIP events around each kernel cost ~10 us of a 0.25 ms step: they are recorded on every TIMING_EVERY-th
    # step of the timed region, which is what the roofline's average launch duration is taken from
    TIMING_EVERY = 8

    # the per-step rate vectors are prepared ahead of the timed region (an optimiser hands them over ready-made)
    rates = [node_rates(s) for s in range(args.warmup + args.steps)]

    def one_step(step, timed=False):
        nl, nm = rates[step]
        timed = timed and rank == 0
        eng.enable_timing(timed)
        if not multi:
            score, fz = eng.get_posterior(nl, nm, prior)
            if timed:
                kernel_ms.append(eng.last_kernel_ms())
            return score
        eng.eval_posterior_async(nl, nm, prior, p_chunks, p_fz)
        # the one exchange step: a single RCCL all_gather of (chunk sums, first-zero index) per rank
        score, fz = D.exchange_packed(dist, torch, packed, gathered, n_chunks, bounds, gathered_host, engine=eng)
        if timed:
            kernel_ms.append(eng.last_kernel_ms())  # the exchange has synchronised the stream
        return score

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # engine priming (not part of the W warm-up steps the caller asked for): the first launches after set-up
    # allocate the scratch buffers, measure the K2 wave grids (~15 evaluations) and run while the GPU is still
    # leaving its idle power state -- at least 30 evaluations and at least 0.25 s of them
    # (with several ranks every step contains a collective, so the count must be the same everywhere: fixed)
    PRIMING = 0
    t_prime = time.perf_counter()
    while (PRIMING < 1000) if multi else (PRIMING < 30 or time.perf_counter() - t_prime < 0.25):
        one_step(PRIMING % max(1, args.warmup + args.steps))
        PRIMING += 1
    last = None
    for s in range(args.warmup):
        last = one_step(s)
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        last = one_step(args.warmup + s, timed=(s % TIMING_EVERY == 0))
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    total_families = F_local * world
    value = total_families * args.steps / dt
    out = {
        "metric": "family-likelihood evals/sec (full tree)",
        "value": value,
        "unit": "family-evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "priming_steps": PRIMING,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[1]: %s; %d families per GPU, R=%d root sizes, %dx%d matrices, "
                        "%d edges, one objective evaluation (matrix build + pruning + posterior + score) per step"
                        % (cfg["desc"], F_local, R, C, C, 2 * tree.n_leaves - 2),
            "families_per_gpu": F_local,
            "n_taxa": cfg["n_taxa"],
            "max_family_size": cfg["m"],
            "parallelism": "families sharded x%d" % world,
            "last_score": last,
        },
    }

    if rank == 0 and kernel_ms:
        km = np.array(kernel_ms)  # columns: K1 matrix build, K2 pruning+posterior, K3 score
        k2_ms = float(km[:, 1].mean())
        b_alg = algorithmic_bytes_per_family(tree.n_leaves, R, C)
        f_alg = b_alg / 4.0  # SURVEY.md 8(d) F_alg: 2 flops per 8-byte matrix element of the dense per-edge product
        # flops the GEMM formulation really issues: only internal child edges are products
        n_int_edges = sum(1 for v in range(tree.n_nodes) if tree.left[v] >= 0 and v != tree.root)
        root_int = sum(1 for ch in (tree.left[tree.root], tree.right[tree.root]) if tree.left[ch] >= 0)
        f_exec = 2.0 * C * (R * root_int + C * (n_int_edges - root_int))
        achieved = f_alg * F_local / (k2_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma",
            "kernel": "k2_prune_mfma4 (v_mfma_f64_4x4x4_4b)" if "mfma4x4" in eng.describe() else
                      "k2_prune_mfma (v_mfma_f64_16x16x4)",
            "kernel_does": "pruning of all families + posterior in one launch",
            "achieved": achieved,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_PEAK_TFLOPS,
            "traffic": pmc_traffic_bytes(args.config, F_local),
            "traffic_unit": "bytes per launch (HBM-side, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, "
                            "profiles/r01_pmc_traffic.json)",
            "algorithmic_flops_per_family": f_alg,
            "families_per_launch": F_local,
            "avg_launch_ms": k2_ms,
            "executed": {"flops_per_family": f_exec, "TFLOP/s": f_exec * F_local / (k2_ms * 1e-3) / 1e12,
                         "frac_of_spec_peak": f_exec * F_local / (k2_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                         "frac_of_measured_16x16x4_issue_ceiling_47.7": f_exec * F_local / (k2_ms * 1e-3) / 1e12 / 47.7,
                         "frac_of_measured_4x4x4_issue_ceiling_73.9": f_exec * F_local / (k2_ms * 1e-3) / 1e12 / 73.9},
            "note": "achieved credits SURVEY.md 8(d) F_alg = the reference's dense product on EVERY child edge "
                    "(what the CPU path executes); 'executed' counts only the products the GEMM formulation issues "
                    "(one-hot leaf edges are column gathers). Spec peak 78.6 TFLOP/s FP64; register-only issue-rate "
                    "ceilings measured on this chip: 47.7 TFLOP/s for v_mfma_f64_16x16x4, 73.9 for "
                    "v_mfma_f64_4x4x4_4b (profiles/r01_mfma_f64_probe.txt, r01_mfma_f64_4x4x4_probe.txt).",
        }
        out["roofline_hbm_effective"] = {
            "achieved": b_alg * F_local / (k2_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": b_alg * F_local / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_family": b_alg,
            "note": "SURVEY.md 8(d) B_alg: bytes the reference's per-family mat-vecs would stream; an EFFECTIVE "
                    "bandwidth (> HBM peak) because the matrices are shared by all families and stay in L2/MALL",
        }
        out["kernel_ms"] = {"k1_matrix_build": float(km[:, 0].mean()), "k2_prune": k2_ms,
                            "k3_score": float(km[:, 2].mean())}
        out["engine"] = eng.describe()

    if rank == 0 and world == 1 and not args.no_search:
        out["lambda_search"] = lambda_search_wallclock(newick, counts, tree, has_mu)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(newick, counts, rng, prior, cfg, eng, tree)

    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if multi:
        dist.destroy_process_group()


def pmc_traffic_bytes(config, families):
    """HBM-side bytes per K2 launch from the committed rocprofv3 PMC passes (collected in their own runs,
    as the counters cannot be read from inside bench.py); None when the recorded workload differs."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if rec["workload"] != config or rec["families_per_launch"] != families:
            return None
        k = rec["kernels"]["k2_prune_mfma"]
        return (k["fetch_kib_corrected"] + k["write_kib"]) * 1024.0
    except Exception:
        return None


def lambda_search_wallclock(newick, counts, tree, has_mu):
    """Second metric of BASELINE.json: wall-clock of the complete `lambda -s` (or `lambdamu -s`) command on
    the bench table through the host driver -- prior fit + Nelder-Mead, every objective call on the GPU
    (cafe/lambda.cpp:369-515)."""
    import tempfile
    from cafe_amd.shell import CafeShell
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "families.tab")
        with open(path, "w") as f:
            f.write("Desc\tFamily ID\t" + "\t".join(tree.leaf_names) + "\n")
            for i, row in enumerate(counts):
                f.write("NA\tF%06d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")
        sh = CafeShell(0, os.path.join(d, "log.txt"))
        sh.dispatch("seed 10")
        sh.dispatch("tree " + newick)
        sh.dispatch("load -i " + path)
        t0 = time.perf_counter()
        sh.dispatch("lambdamu -s" if has_mu else "lambda -s")
        wall = time.perf_counter() - t0
        res = {"command": "lambdamu -s" if has_mu else "lambda -s", "wall_s": wall,
               "search_s": sh.search_seconds, "iterations": sh.iterations, "evaluations": sh.evaluations,
               "fitted": [float(x) for x in sh.params], "score": sh.score, "poisson_lambda": sh.poisson_lambda}
        sh.close()
    return res


def cpu_baseline(newick, counts, rng, prior, cfg, eng, tree):
    """The oracle (CPU restatement of the reference algorithm, dense mat-vec on every edge) timed
    on this box's host cores on a bounded sample of the same table; also used to cross-check the
    GPU values of that sample."""
    from tests import _orc as O
    t = O.PyTree(newick)
    orng = O.make_range(rng.min, rng.max, rng.root_min, rng.root_max)
    lam = np.full(t.n_nodes, cfg["lam"])
    mu = np.full(t.n_nodes, cfg["mu"] if cfg["mu"] >= 0 else -1.0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    probe = counts[:64]
    t0 = time.perf_counter()
    O.eval_posterior(t, probe, orng, lam, mu, prior, nthreads=1)
    t1 = time.perf_counter() - t0
    per_fam_1t = t1 / len(probe)
    n_1t = int(max(64, min(len(counts), 4.0 / per_fam_1t)))
    t0 = time.perf_counter()
    O.eval_posterior(t, counts[:n_1t], orng, lam, mu, prior, nthreads=1)
    rate_1t = n_1t / (time.perf_counter() - t0)
    # the box may expose more hardware threads than its CPU quota: pick the team size that is fastest
    best_threads, best_rate = 1, rate_1t
    n_try = int(max(256, min(len(counts), 2000)))
    for nt in sorted({avail, 128, 64, 32, 16, 8}, reverse=True):
        if nt > avail or nt < 2:
            continue
        O.eval_posterior(t, counts[:256], orng, lam, mu, prior, nthreads=nt)  # spin the team up, untimed
        t0 = time.perf_counter()
        O.eval_posterior(t, counts[:n_try], orng, lam, mu, prior, nthreads=nt)
        r = n_try / (time.perf_counter() - t0)
        if r > best_rate:
            best_threads, best_rate = nt, r
    cores = best_threads
    n_mt = int(max(256, min(len(counts), 6.0 * best_rate)))
    rate_mt = 0.0
    for _ in range(2):
        t0 = time.perf_counter()
        so, fzo, mlo, amo, mpo = O.eval_posterior(t, counts[:n_mt], orng, lam, mu, prior, nthreads=cores)
        rate_mt = max(rate_mt, n_mt / (time.perf_counter() - t0))
    # parity of the same sample on the GPU
    eng.set_families(counts[:n_mt], rng)
    sg, fzg, mlg, amg, mpg = eng.get_posterior(lam, mu, prior, per_family=True)
    rel = float(np.max(np.abs(np.log(mpg) - np.log(mpo)) / np.abs(np.log(mpo))))
    return {
        "value": rate_mt,
        "unit": "family-evals/s",
        "cores": cores,
        "kind": "port",
        "sample": "one objective evaluation of the first %d families of the bench table, OpenMP over "
                  "families on %d threads (best of team sizes <= %d hardware threads; includes the matrix "
                  "build)" % (n_mt, cores, avail),
        "single_thread_value": rate_1t,
        "single_thread_sample": "first %d families, 1 thread" % n_1t,
        "gpu_vs_oracle_max_rel_err_log_posterior": rel,
    }


if __name__ == "__main__":
    main()

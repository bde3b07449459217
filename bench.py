#!/usr/bin/env python
"""bench.py -- family-likelihood evaluations per second of the HIP hot path.

A "step" is ONE objective evaluation (reset_birthdeath_cache + get_posterior,
cafe/cafe_main.c:319-326 + cafe/lambda.cpp:691-724) over the whole count table: every step uses
different rates (as the optimiser does), builds all transition matrices, prunes every family,
reduces the score and returns it to the host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5] [--scaling weak|strong]

Default workload = BASELINE.json configs[1] (cfg2: 10k families, 16 taxa, single lambda).  The other
configs: cfg3 = configs[2] (100k families, 32 taxa, lambda/mu), cfg4 = one GPU's shard of configs[3]
(62,500 of 500k families, 64 taxa, 3 lambda classes by clade), cfg5 = configs[4] (100k families, error
model on every leaf; adds the Monte-Carlo-null launch, R x 1000 simulated families, with its own roofline).

--gpus N > 1: one process per GPU.  Started as a plain `python bench.py --gpus N` the script re-executes
itself under torch.distributed.run with N ranks; started by torch.distributed.run it takes its rank from
the environment.  Families are sharded (weak: the config's table per GPU; strong: one table split N ways),
one RCCL all_gather of the packed per-chunk partial sums per step.
"""
import argparse
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured float4 copy)
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix = vector peak: 256 CUs x 4 SIMDs x 32 flop/cycle x 2.4 GHz
MIN_KERNEL_SAMPLES = 16
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")


def algorithmic_elements_per_family(n_leaves, R, C):
    """SURVEY.md section 8(d) N_el: the reference runs a dense sub-matrix product on every child edge:
    2*R*C + (E-2)*C*C matrix elements per family evaluation, E = 2n-2 edges."""
    E = 2 * n_leaves - 2
    return 2.0 * R * C + (E - 2.0) * C * C


def issued_mfma_flops_per_family(tree, R, C, root_rows=None):
    """Flops of the matrix instructions K2 ISSUES per family, tile padding included: only internal child
    edges are products (one-hot leaf edges are column gathers); a product covers roundup16(rows) x roundup4(C)
    (16-row tiles, 4-deep k-steps).  Returns (issued incl. padding, useful = exact rows x C)."""
    n_int_edges = sum(1 for v in range(tree.n_nodes) if tree.left[v] >= 0 and v != tree.root)
    root_int = sum(1 for ch in (tree.left[tree.root], tree.right[tree.root]) if tree.left[ch] >= 0)
    rr = R if root_rows is None else root_rows
    pad16 = lambda x: 16 * ((x + 15) // 16)
    kpad = 4 * ((C + 3) // 4)
    issued = 2.0 * kpad * (pad16(rr) * root_int + pad16(C) * (n_int_edges - root_int))
    useful = 2.0 * C * (rr * root_int + C * (n_int_edges - root_int))
    return issued, useful


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, N ranks."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--families", type=int, default=None, help="families per GPU (default: the config's)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the table per GPU is fixed; strong: ONE table (the config's F, 500k for cfg4) split over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="skip the lambda-search wall-clock leg")
    ap.add_argument("--no-probes", action="store_true", help="skip the measured HBM / MFMA ceilings")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --same-device)")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: take the multi-rank code path (process group + packed all_gather) even with 1 rank")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: all ranks share GPU 0 (functional check of the N>1 path on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # the contract is ONE JSON line on stdout: whatever libraries print to file descriptor 1 meanwhile (the host
    # driver echoes some commands with printf) is sent to stderr, and the line is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    backend = args.backend or ("gloo" if args.same_device else "nccl")
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import cafe_amd
    from cafe_amd import distributed as D
    from cafe_amd import prior as cprior
    from cafe_amd import synth
    from cafe_amd import tree as ctree

    # ---- workload: this rank's shard of the family table -------------------------------
    cfg = dict(synth.CONFIGS[args.config])
    per_gpu_default = 62500 if args.config == "cfg4" else cfg["F"]   # configs[3] is quoted on 8 GPUs
    if args.scaling == "strong":
        F_total = args.families or cfg["F"]
        b = D.shard_bounds(F_total, world)
        F_local = b[rank][1] - b[rank][0]
    else:
        F_local = args.families or per_gpu_default
        F_total = F_local * world
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"]))
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
    if cfg.get("error_model"):
        eng.set_error_model(synth.banded_error_matrix(rng.max))
    n_chunks = eng.num_chunks()
    eng.enable_timing(False)

    if args.scaling == "strong":
        bounds = D.shard_bounds(F_total, world)
    else:
        bounds = [(r * F_local, (r + 1) * F_local) for r in range(world)]
    slots = max(1, max((hi - lo + D.CHUNK - 1) // D.CHUNK for lo, hi in bounds))
    packed, p_chunks, p_fz = D.packed_buffer(torch, slots, "cuda")
    gathered = torch.zeros((slots + 1) * world, dtype=torch.float64, device="cuda")
    gathered_host = torch.zeros((slots + 1) * world, dtype=torch.float64).pin_memory()

    def node_rates(step):
        return synth.node_rates(tree, cfg, 1.0 + 0.003 * (step % 97), 1.0 + 0.002 * (step % 89))

    kernel_ms = []
    exchange_s = [0.0, 0]

    # HIP events around each kernel cost ~10 us of a step: they are recorded on every TIMING_EVERY-th step of the
    # timed region and, if that leaves fewer than MIN_KERNEL_SAMPLES, on extra steps run right after it
    TIMING_EVERY = 8

    # the per-step rate vectors are prepared ahead of the timed region (an optimiser hands them over ready-made)
    n_extra = MIN_KERNEL_SAMPLES
    rates = [node_rates(s) for s in range(args.warmup + args.steps + n_extra)]

    def one_step(step, timed=False):
        nl, nm = rates[step % len(rates)]
        timed = timed and rank == 0
        eng.enable_timing(timed)
        if not multi:
            score, fz = eng.get_posterior(nl, nm, prior)
            if timed:
                kernel_ms.append(list(eng.last_kernel_ms()) + [eng.last_tables_ms()])
            return score
        eng.eval_posterior_async(nl, nm, prior, p_chunks, p_fz)
        # the one exchange step: a single RCCL all_gather of (chunk sums, first-zero index) per rank
        t_x = time.perf_counter()
        score, fz = D.exchange_packed(dist, torch, packed, gathered, slots, bounds, gathered_host, engine=eng)
        exchange_s[0] += time.perf_counter() - t_x
        exchange_s[1] += 1
        if timed:
            kernel_ms.append(list(eng.last_kernel_ms()) + [eng.last_tables_ms()])  # the exchange has synchronised the stream
        return score

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # engine priming (not part of the W warm-up steps the caller asked for): the first launches after set-up
    # allocate the scratch buffers, measure the K2 wave grids (up to ~30 evaluations) and run while the GPU is still
    # leaving its idle power state -- at least 40 evaluations and at least 0.25 s of them
    # (with several ranks every step contains a collective, so the count must be the same everywhere: fixed)
    PRIMING = 0
    t_prime = time.perf_counter()
    n_prime_multi = 1000 if F_local <= 20000 else 60
    while (PRIMING < n_prime_multi) if multi else (PRIMING < 40 or time.perf_counter() - t_prime < 0.25):
        one_step(PRIMING)
        PRIMING += 1
    last = None
    for s in range(args.warmup):
        last = one_step(s)
    kernel_ms.clear()
    exchange_s[0], exchange_s[1] = 0.0, 0
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        last = one_step(args.warmup + s, timed=(s % TIMING_EVERY == 0))
    barrier()
    dt = time.perf_counter() - t0
    exchange_ms = 1000.0 * exchange_s[0] / max(exchange_s[1], 1)
    samples_in_region = len(kernel_ms)
    extra = 0
    while rank == 0 and len(kernel_ms) < MIN_KERNEL_SAMPLES and not multi:
        one_step(args.warmup + args.steps + extra, timed=True)
        extra += 1
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    total_families = F_total
    value = total_families * args.steps / dt
    idx = cfg["baseline_index"]
    shard_note = " (one GPU's shard of the 500k-family table)" if args.config == "cfg4" and args.scaling == "weak" else ""
    out = {
        "metric": "family-likelihood evals/sec (full tree)",
        "value": value,
        "unit": "family-evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[%d]%s: %s; %d families per GPU, R=%d root sizes, %dx%d matrices, "
                        "%d edges, one objective evaluation (matrix build + pruning + posterior + score) per step"
                        % (idx, shard_note, cfg["desc"], F_local, R, C, C, 2 * tree.n_leaves - 2),
            "baseline_config_index": idx,
            "families_per_gpu": F_local,
            "families_total": F_total,
            "n_taxa": cfg["n_taxa"],
            "max_family_size": cfg["m"],
            "parallelism": "families sharded x%d" % world,
            "priming_steps_before_warmup": PRIMING,
            "priming_note": "untimed evaluations before the W warm-up steps: scratch allocation, the library's "
                            "measured choice of the K2 wave grid (~15-20 ordinary evaluations) and the clock ramp",
            "last_score": last,
        },
    }
    if multi:
        out["rccl_ranks"] = dist.get_world_size()
        out["backend"] = backend
        out["exchange_ms_per_step"] = exchange_ms

    if rank == 0 and kernel_ms:
        km = np.array(kernel_ms)  # columns: K1 matrix build, K2 pruning+posterior (tables + walk), K3 score, tables alone
        k2_ms = float(km[:, 1].mean())
        tables_ms = float(km[:, 3].mean())
        walk_ms = k2_ms - tables_ms
        desc = eng.describe()
        nf = int(re.search(r"NF=(\d+)", desc).group(1))
        grid = (F_local + nf - 1) // nf
        n_el = algorithmic_elements_per_family(tree.n_leaves, R, C)
        issued, useful = issued_mfma_flops_per_family(tree, R, C)
        # every workgroup issues the matrix instructions of NF family slots, filled or not.  The library reports
        # what the last evaluation issued: the family walk (of the REDUCED tree when the table compresses) and the
        # factor tables of the compressed subtrees (cafehip_last_issued_flops); without compression the walk figure
        # must equal the tree formula above
        walk_fl, table_fl = eng.last_issued_flops()
        compressed = table_fl > 0
        if not compressed and abs(walk_fl - issued * grid * nf) > 1e-9 * walk_fl:
            raise SystemExit("issued-flop accounting: library %.6g vs tree formula %.6g" % (walk_fl, issued * grid * nf))
        achieved = walk_fl / (walk_ms * 1e-3) / 1e12
        frac = achieved / FP64_PEAK_TFLOPS
        total_frac = (walk_fl + table_fl) / (k2_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
        if not (frac <= 1.0 and total_frac <= 1.0):
            raise SystemExit("roofline fraction %.3f / %.3f > 1: the flop accounting is wrong" % (frac, total_frac))
        out["roofline"] = {
            "bound": "mfma",
            "kernel": "k2_prune_mfma4 (v_mfma_f64_4x4x4_4b)" if "mfma4x4" in desc else "k2_prune_mfma (v_mfma_f64_16x16x4)",
            "kernel_does": "the family walk: pruning of all families + posterior in one launch" +
                           (" over the REDUCED tree (compressed subtrees are row gathers from factor tables built by "
                            "the k2c_nodes launches just before it: see factor_tables / pruning_total)" if compressed else ""),
            "achieved": achieved,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": frac,
            "flops_counted": "matrix-instruction flops ISSUED by the launch, tile padding included: products on internal "
                             "child edges only (one-hot leaf edges and compressed subtrees are gathers), roundup16(rows) x "
                             "roundup4(C) per product, NF family slots per workgroup x %d workgroups" % grid,
            "issued_flops_per_launch": walk_fl,
            "avg_launch_ms": walk_ms,
            "launch_samples": len(kernel_ms),
            "launch_samples_in_timed_region": samples_in_region,
            "min_launch_ms": float((km[:, 1] - km[:, 3]).min()),
            "max_launch_ms": float((km[:, 1] - km[:, 3]).max()),
            "families_per_launch": F_local,
            "factor_tables": None if not compressed else {
                "kernel": "k2c_nodes (v_mfma_f64_16x16x4): one launch per compression level, 16 states per workgroup",
                "launches_per_evaluation": int(re.search(r"levels=(\d+)", desc).group(1)),
                "states": int(re.search(r"states=(\d+)", desc).group(1)),
                "ms_per_evaluation": tables_ms,
                "issued_flops": table_fl,
                "achieved_TFLOP/s": table_fl / (tables_ms * 1e-3) / 1e12,
                "frac": table_fl / (tables_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            },
            "pruning_total": {
                "what": "all pruning launches of one evaluation (factor tables + walk), HIP events around them",
                "ms": k2_ms,
                "issued_flops": walk_fl + table_fl,
                "achieved_TFLOP/s": (walk_fl + table_fl) / (k2_ms * 1e-3) / 1e12,
                "frac": total_frac,
                "uncompressed_walk_would_issue": issued * grid * nf,
                "work_saved_by_subtree_state_compression": 1.0 - (walk_fl + table_fl) / (issued * grid * nf),
            },
            "useful_flops_per_launch": useful * F_local,
            "useful_flops_note": "exact rows x C for every internal edge and family, i.e. what an uncompressed walk without "
                                 "tile padding executes; with compression fewer are executed, so the two rates below are "
                                 "NOT utilisations",
            "useful_TFLOP/s": useful * F_local / (k2_ms * 1e-3) / 1e12,
            "useful_frac": useful * F_local / (k2_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
        }
        out["roofline"].update(pmc_traffic(args.config, F_local))
        # SURVEY.md 8(d)'s reference-faithful accounting (a dense product on EVERY child edge, as the CPU path
        # executes it): kept for continuity, NOT a utilisation -- most of these flops/bytes are never issued/moved
        out["algorithmic_credit"] = {
            "what": "SURVEY.md 8(d) F_alg = 2*N_el flops and B_alg = 8*N_el bytes per family evaluation: what the "
                    "reference's per-family dense mat-vecs execute/stream; a rate comparable with the CPU path, "
                    "not a fraction of any hardware peak",
            "N_el_per_family": n_el,
            "F_alg_TFLOP/s": 2.0 * n_el * F_local / (k2_ms * 1e-3) / 1e12,
            "B_alg_effective_GB/s": 8.0 * n_el * F_local / (k2_ms * 1e-3) / 1e9,
        }
        out["kernel_ms"] = {"k1_matrix_build": float(km[:, 0].mean()), "k2_prune": k2_ms,
                            "k3_score": float(km[:, 2].mean())}
        out["engine"] = desc

    if rank == 0 and not args.no_probes:
        out["roofline_measured_peaks"] = measured_peaks(local_rank)
        if "roofline" in out and out["roofline_measured_peaks"].get("mfma_f64_4x4x4_TFLOP/s"):
            mp = out["roofline_measured_peaks"]
            key = "mfma_f64_4x4x4_TFLOP/s" if "mfma4x4" in out["engine"] else "mfma_f64_16x16x4_TFLOP/s"
            out["roofline"]["frac_of_measured_register_only_ceiling"] = out["roofline"]["achieved"] / mp[key]

    if rank == 0 and world == 1 and args.config == "cfg5":
        out["mc_null"] = mc_null_leg(eng, tree, cfg, rng, torch)

    if rank == 0 and world == 1 and not args.no_search:
        out["lambda_search"] = lambda_search_wallclock(newick, counts, tree, cfg, rng)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(newick, counts, rng, prior, cfg, eng, tree)

    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    eng.close()
    if multi:
        dist.destroy_process_group()


def pmc_traffic(config, families, kernel="k2"):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (collected in their own runs -- the
    counters cannot be read from inside bench.py -- by tools/collect_pmc.py; FETCH_SIZE x2 gfx950 correction,
    /opt/skills/guides/MI355X_MICROARCH.md section HBM)."""
    try:
        rec = json.load(open(TRAFFIC_FILE))["%s:%d:%s" % (config, families, kernel)]
        return {"traffic": rec["traffic_bytes"], "traffic_unit": "bytes per launch, HBM side: 2 x FETCH_SIZE + WRITE_SIZE",
                "factor_tables_traffic": rec.get("tables_traffic_bytes"),
                "traffic_source": os.path.relpath(TRAFFIC_FILE, ROOT) + " (" + rec.get("note", "") + ")",
                "traffic_minimal_bytes": rec.get("minimal_bytes")}
    except Exception:
        return {"traffic": None, "traffic_source": "no PMC record for %s with %d families per launch" % (config, families)}


def measured_peaks(device):
    """Ceilings measured on THIS chip by cafe_amd/csrc/probe.hip (separate library, measurement only)."""
    import ctypes as C
    from cafe_amd import build as B
    try:
        L = C.CDLL(B.PROBE_LIB)
    except OSError as e:
        return {"error": str(e)}
    out = {}
    v = C.c_double()
    for name, fn in (("hbm_triad_GB/s", L.cafeprobe_hbm_triad), ("hbm_copy_GB/s", L.cafeprobe_hbm_copy)):
        fn.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_double)]
        if fn(device, 1 << 30, C.byref(v)) == 0:
            out[name] = v.value
    L.cafeprobe_mfma_f64.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    for shape, key in ((4, "mfma_f64_4x4x4_TFLOP/s"), (16, "mfma_f64_16x16x4_TFLOP/s")):
        best = 0.0
        for wg in (1, 2):
            if L.cafeprobe_mfma_f64(device, shape, wg, C.byref(v)) == 0:
                best = max(best, v.value)
        out[key] = best
    out["note"] = ("stream triad / 16-byte copy over 1 GiB arrays (beyond the 256 MiB Infinity Cache); register-only "
                   "issue rate of the two FP64 matrix instructions, 2-4 waves per SIMD; spec peaks: 8000 GB/s, 78.6 TFLOP/s")
    return out


def mc_null_leg(eng, tree, cfg, rng, torch):
    """BASELINE configs[4] tail: the Monte-Carlo null of the report -- R root sizes x 1000 simulated families
    (get_random_probabilities, cafe/conditional_distribution.cpp:10-44), every one scored with a one-row root
    in ONE batched launch of the pruning kernel (cafehip_eval_root_likelihoods).  Families are simulated here
    with numpy from the device-built matrices (workload generation; the product's host driver draws them in the
    reference's rand() order)."""
    R = rng.root_max - rng.root_min + 1
    C = rng.max + 1
    trials = 1000
    from cafe_amd import synth
    nl, nm = synth.node_rates(tree, cfg)
    eng.reset_birthdeath_cache(nl, nm)
    mats = {v: eng.get_matrix(v) for v in range(tree.n_nodes) if v != tree.root}
    counts, lo, cm = synth.simulate_null_rows(tree, mats, rng, trials, cfg["seed"] + 77)
    B = len(lo)
    ms = []
    eng.enable_timing(True)
    t_wall = []
    for _ in range(5):
        t0 = time.perf_counter()
        like = eng.eval_root_likelihoods(counts, lo, lo, cm)
        t_wall.append(time.perf_counter() - t0)
        ms.append(eng.last_batch_ms())
    eng.enable_timing(False)
    k_ms = float(np.mean(ms[1:]))
    desc = eng.describe()
    nf = int(re.search(r"NF=(\d+)", desc).group(1))
    grid = (B + nf - 1) // nf
    issued, useful = issued_mfma_flops_per_family(tree, R, C)   # the kernel computes all R root rows of a row's tile
    achieved = issued * grid * nf / (k_ms * 1e-3) / 1e12
    frac = achieved / FP64_PEAK_TFLOPS
    if not frac <= 1.0:
        raise SystemExit("MC-null roofline fraction %.3f > 1: the flop accounting is wrong" % frac)
    out = {
        "what": "Monte-Carlo null of the report: %d root sizes x %d simulated families = %d rows, one batched launch "
                "(per-row root size and column limit), no error model (cafe/cafe_tree.c:485-494)" % (R, trials, B),
        "rows": B,
        "launch_ms": k_ms,
        "call_wall_ms_incl_pcie": 1000.0 * float(np.mean(t_wall[1:])),
        "rows_per_s": B / (k_ms * 1e-3),
        "finite_likelihoods": int(np.isfinite(like).sum()),
        "roofline": {"bound": "mfma", "kernel": "k2_prune_mfma (batch mode)", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": frac, "avg_launch_ms": k_ms, "launch_samples": len(ms) - 1,
                     "issued_flops_per_launch": issued * grid * nf},
        "engine": desc,
    }
    out["roofline"].update(pmc_traffic("cfg5", B, "mcnull"))
    return out


def lambda_search_wallclock(newick, counts, tree, cfg, rng):
    """Second metric of BASELINE.json: wall-clock of the complete `lambda -s` (or `lambdamu -s`) command on
    the bench table through the host driver -- prior fit + Nelder-Mead, every objective call on the GPU
    (cafe/lambda.cpp:369-515)."""
    import tempfile
    from cafe_amd import synth
    from cafe_amd.shell import CafeShell
    has_mu = cfg["mu"] >= 0
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
        if cfg.get("error_model"):
            em = os.path.join(d, "errormodel.txt")
            synth.write_error_model_file(em, rng.max)
            sh.dispatch("errormodel -model %s -all" % em)
        if has_mu:
            command = "lambdamu -s"
        elif cfg.get("n_classes"):
            command = "lambda -s -t " + synth.clade_classes(tree, cfg["n_classes"])[1]
        else:
            command = "lambda -s"
        t0 = time.perf_counter()
        sh.dispatch(command)
        wall = time.perf_counter() - t0
        res = {"command": command if len(command) < 40 else command[:24] + "<lambda tree>", "wall_s": wall,
               "search_s": sh.search_seconds, "iterations": sh.iterations, "evaluations": sh.evaluations,
               "fitted": [float(x) for x in sh.params], "score": sh.score, "poisson_lambda": sh.poisson_lambda}
        sh.close()
    return res


def physical_cores():
    """(physical cores, hardware threads) of this box from lscpu."""
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        get = lambda k: int(re.search(r"^%s:\s*(\d+)" % re.escape(k), txt, re.M).group(1))
        return get("Core(s) per socket") * get("Socket(s)"), get("CPU(s)")
    except Exception:
        n = os.cpu_count() or 1
        return n, n


def cpu_baseline(newick, counts, rng, prior, cfg, eng, tree):
    """The oracle (CPU restatement of the reference algorithm, dense mat-vec on every edge) built on THIS box
    with -O3 -march=native and timed on its host cores on a bounded sample of the same table;
    also used to cross-check the GPU values of that sample."""
    os.environ["CAFE_ORACLE_NATIVE"] = "1"   # before the first import of the oracle binding
    from cafe_amd import synth
    from tests import _orc as O
    t = O.PyTree(newick)
    orng = O.make_range(rng.min, rng.max, rng.root_min, rng.root_max)
    lam, mu = synth.node_rates(tree, cfg)
    err = synth.banded_error_matrix(rng.max) if cfg.get("error_model") else None
    ekw = dict(errormatrix=err, err_mfs=rng.max, leaf_has_err=np.ones(t.n_nodes, np.uint8)) if err is not None else {}
    phys, hw = physical_cores()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = hw
    probe = counts[:32]
    t0 = time.perf_counter()
    O.eval_posterior(t, probe, orng, lam, mu, prior, nthreads=1, **ekw)
    per_fam_1t = (time.perf_counter() - t0) / len(probe)
    n_1t = int(max(32, min(len(counts), 4.0 / per_fam_1t)))
    t0 = time.perf_counter()
    O.eval_posterior(t, counts[:n_1t], orng, lam, mu, prior, nthreads=1, **ekw)
    rate_1t = n_1t / (time.perf_counter() - t0)
    # one team per physical core is the stated configuration; the box may cap the process below that
    # (cgroup quota), so smaller teams are tried too and the fastest is reported with its size
    tried = {}
    n_try = int(max(256, min(len(counts), 2.0 * rate_1t * min(phys, avail) * 0.5)))
    for nt in sorted({min(phys, avail), min(hw, avail), 128, 64, 32, 16, 8}, reverse=True):
        if nt > avail or nt < 2:
            continue
        O.eval_posterior(t, counts[:256], orng, lam, mu, prior, nthreads=nt, **ekw)  # spin the team up, untimed
        t0 = time.perf_counter()
        O.eval_posterior(t, counts[:n_try], orng, lam, mu, prior, nthreads=nt, **ekw)
        tried[nt] = n_try / (time.perf_counter() - t0)
    best_threads = max(tried, key=tried.get) if tried else 1
    best_rate = tried.get(best_threads, rate_1t)
    n_mt = int(max(256, min(len(counts), 6.0 * best_rate)))
    rate_mt = 0.0
    for _ in range(2):
        t0 = time.perf_counter()
        so, fzo, mlo, amo, mpo = O.eval_posterior(t, counts[:n_mt], orng, lam, mu, prior, nthreads=best_threads, **ekw)
        rate_mt = max(rate_mt, n_mt / (time.perf_counter() - t0))
    # parity of the same sample on the GPU
    eng.set_families(counts[:n_mt], rng)
    sg, fzg, mlg, amg, mpg = eng.get_posterior(lam, mu, prior, per_family=True)
    rel = float(np.max(np.abs(np.log(mpg) - np.log(mpo)) / np.abs(np.log(mpo))))
    return {
        "value": rate_mt,
        "unit": "family-evals/s",
        "cores": best_threads,
        "physical_cores": phys,
        "hardware_threads": hw,
        "threads_available_to_this_process": avail,
        "kind": "port",
        "build": "oracle/cafe_oracle.c, gcc -O3 -march=native -ffp-contract=off -fopenmp, built on this box",
        "sample": "one objective evaluation of the first %d families of the bench table, OpenMP over families on "
                  "%d threads (fastest of the team sizes tried: %s; includes the matrix build)"
                  % (n_mt, best_threads, ", ".join("%d: %.0f/s" % kv for kv in sorted(tried.items()))),
        "single_thread_value": rate_1t,
        "single_thread_sample": "first %d families, 1 thread" % n_1t,
        "gpu_vs_oracle_max_rel_err_log_posterior": rel,
    }


if __name__ == "__main__":
    main()

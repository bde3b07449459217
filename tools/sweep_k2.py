#!/usr/bin/env python
"""Sweep the K2 wave-grid configurations on one workload and print kernel times (GPU box)."""
import os, sys, itertools, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cafe_amd
from cafe_amd import synth, prior as cprior, tree as ctree

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
cfg = dict(synth.CONFIGS[name])
F = F or cfg["F"]
newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
tree = ctree.CafeTree(newick)
counts = synth.simulate_families(tree, F, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
rng = cafe_amd.init_family_size(cfg["m"])
prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
eng = cafe_amd.Engine(0)
tree.apply(eng)
eng.set_families(counts, rng)
eng.enable_timing(True)
nl = np.full(tree.n_nodes, cfg["lam"]); nm = np.full(tree.n_nodes, cfg["mu"])
RT = (rng.max + 1 + 15) // 16
ref = None
rows = []
cands = [None]
for nft in (1, 2):
    for wr in (1, 2, 4, 8):
        nrt = -(-RT // wr)
        if nrt > 7 or nft * nrt > 8: continue
        for wf in (1, 2, 4, 8):
            if wf * wr > 8: continue
            cands.append((nft, nrt, wf, wr))
for c in cands:
    if c is None: os.environ.pop("CAFEHIP_K2CFG", None)
    else: os.environ["CAFEHIP_K2CFG"] = "%d,%d,%d,%d" % c
    try:
        ms = []
        for it in range(6):
            score, fz = eng.get_posterior(nl * (1 + 0.001 * it), nm, prior)
            if it >= 2: ms.append(eng.last_kernel_ms())
        ms = np.array(ms).mean(axis=0)
        s0, _ = eng.get_posterior(nl, nm, prior)
        if ref is None: ref = s0
        print("%-12s k1 %.3f  k2 %.3f  k3 %.3f ms  score_diff %.3g  %s" % (c, ms[0], ms[1], ms[2], s0 - ref, eng.describe().split("k2:")[1]), flush=True)
    except Exception as e:
        print(c, "FAILED", e, flush=True)

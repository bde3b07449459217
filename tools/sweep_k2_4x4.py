#!/usr/bin/env python
"""Sweep the 4-family-granularity K2 (4x4x4 MFMA) wave grids on one workload (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cafe_amd
from cafe_amd import synth, prior as cprior, tree as ctree
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
cfg = dict(synth.CONFIGS[name]); F = F or cfg["F"]
newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
tree = ctree.CafeTree(newick)
counts = synth.simulate_families(tree, F, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
rng = cafe_amd.init_family_size(cfg["m"])
prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
eng = cafe_amd.Engine(0); tree.apply(eng); eng.set_families(counts, rng); eng.enable_timing(True)
nl = np.full(tree.n_nodes, cfg["lam"]); nm = np.full(tree.n_nodes, cfg["mu"])
RT = (rng.max + 1 + 15) // 16
def run(tag):
    ms = []
    for it in range(7):
        s, _ = eng.get_posterior(nl * (1 + 0.001 * it), nm, prior)
        if it >= 3: ms.append(eng.last_kernel_ms()[1])
    s0, _ = eng.get_posterior(nl, nm, prior)
    print("%-22s k2 %.3f ms  score %.9f  %s" % (tag, np.mean(ms), s0, eng.describe().split("k2:")[1]), flush=True)
os.environ.pop("CAFEHIP_MFMA", None); run("auto")
os.environ["CAFEHIP_MFMA"] = "16"; run("auto16")
os.environ["CAFEHIP_MFMA"] = "4"; run("auto4")
for wr in (2, 4, 8):
    nrt = -(-RT // wr)
    if nrt > 7: continue
    for G in range(1, 9):
        if G * nrt > 24: continue
        for wf in (1, 2):
            if wf * wr > 8: continue
            os.environ["CAFEHIP_K2CFG4"] = "%d,%d,%d,%d" % (G, nrt, wf, wr)
            try: run("G%d nrt%d wf%d wr%d" % (G, nrt, wf, wr))
            except Exception as e: print("FAILED", G, nrt, wf, wr, e)

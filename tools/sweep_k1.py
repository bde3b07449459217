#!/usr/bin/env python
"""K1 timing vs keys-per-block (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cafe_amd
from cafe_amd import synth, prior as cprior, tree as ctree
for name, F in (("cfg2", 2000), ("cfg3", 2000), ("cfg4", 2000)):
    cfg = dict(synth.CONFIGS[name])
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
    tree = ctree.CafeTree(newick)
    counts = synth.simulate_families(tree, F, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    rng = cafe_amd.init_family_size(cfg["m"])
    prior = cprior.prior_rfsize_poisson(rng.root_min, 8.0)
    eng = cafe_amd.Engine(0)
    tree.apply(eng); eng.set_families(counts, rng); eng.enable_timing(True)
    nl = np.full(tree.n_nodes, cfg["lam"]); nm = np.full(tree.n_nodes, cfg["mu"])
    for kpb in (None, 1, 2, 3, 4, 6, 8, 12):
        if kpb is None: os.environ.pop("CAFEHIP_K1KPB", None)
        else: os.environ["CAFEHIP_K1KPB"] = str(kpb)
        ms = []
        for it in range(8):
            eng.get_posterior(nl * (1 + 0.001 * it), nm, prior)
            if it >= 3: ms.append(eng.last_kernel_ms()[0])
        print(name, "kpb", kpb, "k1 %.4f ms" % np.mean(ms), flush=True)
    eng.close()

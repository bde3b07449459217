import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import cafe_amd
from cafe_amd import synth, prior as cprior, tree as ctree
for name, F in (("cfg2", 10000), ("cfg2", 3000), ("cfg2", 40000), ("cfg3", 20000), ("cfg4", 62464)):
    cfg = dict(synth.CONFIGS[name])
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
    tree = ctree.CafeTree(newick)
    counts = synth.simulate_families(tree, F, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    rng = cafe_amd.init_family_size(cfg["m"])
    prior = cprior.prior_rfsize_poisson(rng.root_min, 8.0)
    eng = cafe_amd.Engine(0); tree.apply(eng); eng.set_families(counts, rng); eng.enable_timing(True)
    nl = np.full(tree.n_nodes, cfg["lam"]); nm = np.full(tree.n_nodes, cfg["mu"])
    for mode in (None, "16", "4"):
        if mode is None: os.environ.pop("CAFEHIP_MFMA", None)
        else: os.environ["CAFEHIP_MFMA"] = mode
        ms = []
        for it in range(7):
            eng.get_posterior(nl * (1 + 0.001 * it), nm, prior)
            if it >= 3: ms.append(eng.last_kernel_ms()[1])
        print(name, F, mode, "k2 %.3f ms" % np.mean(ms), eng.describe().split("k2:")[1], flush=True)
    eng.close()

#!/usr/bin/env python
"""Kernel times of the objective evaluation for a list of workloads with the library named by CAFEHIP_LIB
(or the in-tree one): `python tools/ab_one.py cfg2:10000 cfg3:100000 ...`.  Synthetic tables are cached under
/tmp so that several variants can be compared quickly (tools/ab_variants.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(name, F):
    from cafe_amd import synth
    from cafe_amd.tree import CafeTree
    cfg = dict(synth.CONFIGS[name])
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"]))
    tree = CafeTree(newick)
    path = "/tmp/ab_%s_%d.npy" % (name, F)
    if os.path.exists(path):
        counts = np.load(path)
    else:
        counts = synth.simulate_families(tree, F, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
        np.save(path, counts)
    return tree, counts, cfg


def main():
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import prior as cprior, synth
    for spec in sys.argv[1:]:
        name, F = spec.split(":")
        F = int(F)
        tree, counts, cfg = table(name, F)
        rng = cafe_amd.init_family_size(cfg["m"])
        prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
        eng = cafe_amd.Engine(0)
        tree.apply(eng)
        eng.set_families(counts, rng)
        if cfg.get("error_model"):
            eng.set_error_model(synth.banded_error_matrix(rng.max))
        nl, nm = synth.node_rates(tree, cfg)
        t0 = time.perf_counter()
        n = 0
        while n < 40 or (time.perf_counter() - t0 < 0.3 and n < 400):
            score, fz = eng.get_posterior(nl, nm, prior)
            n += 1
        eng.enable_timing(True)
        ks = []
        for _ in range(16):
            score, fz = eng.get_posterior(nl, nm, prior)
            ks.append(eng.last_kernel_ms() + [eng.last_tables_ms()])
        eng.enable_timing(False)
        t0 = time.perf_counter()
        for _ in range(20):
            eng.get_posterior(nl, nm, prior)
        step = (time.perf_counter() - t0) / 20 * 1e3
        ks = np.array(ks)
        d = eng.describe()
        print("%-5s F=%-7d k1 %.4f  k2 mean %.4f min %.4f (tables %.4f)  k3 %.4f  step %.4f ms  score %.9f  %s" % (
            name, F, ks[:, 0].mean(), ks[:, 1].mean(), ks[:, 1].min(), ks[:, 3].mean(), ks[:, 2].mean(), step, score,
            d[d.index("k2:"):]), flush=True)
        eng.close()


if __name__ == "__main__":
    main()

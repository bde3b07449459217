#!/usr/bin/env python
"""The Monte-Carlo-null launch of BASELINE configs[4] on its own (for profiling): the table first (45 evaluations, no
error model), then R x 1000 simulated rows through cafehip_eval_root_likelihoods, `reps` times.
Usage: python tools/mcnull_one.py [reps] [16:nftw,nrtw,wf,wr | 4:G,nrtw,wf,wr | key=value[;key=value] ...]
(pinned wave grids / options of cafehip_set_option, applied cumulatively; one timing line per argument)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth
    from cafe_amd import prior as cprior
    tree, counts, cfg = synth.make_config("cfg5")
    rng = cafe_amd.init_family_size(cfg["m"])
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    lam, mu = synth.node_rates(tree, cfg)
    # as in a session: the lambda search has been through the table (and settled the K2 wave grid) before the report
    prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
    for _ in range(45):
        eng.get_posterior(lam, mu, prior)
    eng.reset_birthdeath_cache(lam, mu)
    mats = {v: eng.get_matrix(v) for v in range(tree.n_nodes) if v != tree.root}
    rows, lo, cm = synth.simulate_null_rows(tree, mats, rng, 1000, cfg["seed"] + 77)
    eng.enable_timing(True)
    ref = None
    for grid in [None] + sys.argv[2:]:
        if grid and "=" in grid:
            for kv in grid.split(";"):
                eng.set_option(*kv.split("=", 1))
        elif grid:
            shape, cfg_ = grid.split(":")
            eng.set_option("mfma", shape)
            eng.set_option("k2cfg" if shape == "16" else "k2cfg4", cfg_)
        ms = []
        try:
            for _ in range(reps):
                like = eng.eval_root_likelihoods(rows, lo, lo, cm)
                ms.append(eng.last_batch_ms())
        except Exception as e:
            print("mcnull %s: %s" % (grid, e))
            continue
        if ref is None:
            ref = like
        same = all(np.array_equal(a, b) for a, b in zip(ref, like)) if isinstance(like, (list, tuple)) else np.array_equal(ref, like)
        d = eng.describe()
        print("mcnull %s rows %d  launch ms %s  same=%s  %s" % (grid or "default", len(lo), " ".join("%.3f" % x for x in ms), same,
                                                              d if grid is None else d[d.index("k2:"):d.index("grid=") + 12]))
    eng.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The Monte-Carlo-null launch of BASELINE configs[4] on its own (for profiling): the table first (45 evaluations, no
error model), then R x 1000 simulated rows through cafehip_eval_root_likelihoods, `reps` times.  Usage: python tools/mcnull_one.py [reps]"""
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
    ms = []
    for _ in range(reps):
        like = eng.eval_root_likelihoods(rows, lo, lo, cm)
        ms.append(eng.last_batch_ms())
    print("mcnull rows %d  launch ms %s  %s" % (len(lo), " ".join("%.3f" % x for x in ms), eng.describe()))
    eng.close()


if __name__ == "__main__":
    main()

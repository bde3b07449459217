#!/usr/bin/env python
"""Kernel times of one objective evaluation on the reference's shipped example (59 families, 5 taxa), and its `lambda -s`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import cafe_amd
from cafe_amd import prior as cprior
from cafe_amd.tree import CafeTree
tree = CafeTree("(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)")
rows = []
with open(os.path.join(ROOT, "tests", "golden", "example_data.tab")) as f:
    header = f.readline().rstrip("\n").split("\t")
    names = [h.lower() for h in header[2:]]
    col = [names.index(n.lower()) for n in tree.leaf_names]
    for line in f:
        p = line.rstrip("\n").split("\t")
        if len(p) > 2:
            rows.append([int(p[2 + c]) for c in col])
counts = np.array(rows, np.int32)
rng = cafe_amd.init_family_size(int(counts.max()))
prior = cprior.prior_rfsize_poisson(rng.root_min, cprior.poisson_lambda_mle(counts))
for opts in sys.argv[1:] or ["default"]:
    eng = cafe_amd.Engine(0)
    for kv in opts.split(","):
        if kv != "default":
            k, v = kv.split("=")
            eng.set_option(k, v)
    tree.apply(eng)
    eng.set_families(counts, rng)
    nl = np.full(tree.n_nodes, 0.0107)
    nm = np.full(tree.n_nodes, -1.0)
    for _ in range(200):
        eng.get_posterior(nl, nm, prior)
    eng.enable_timing(True)
    ks = []
    for _ in range(16):
        s, fz = eng.get_posterior(nl, nm, prior)
        ks.append(eng.last_kernel_ms())
    eng.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.get_posterior(nl, nm, prior)
    step = (time.perf_counter() - t0) / 200 * 1e3
    ks = np.median(np.array(ks), axis=0)
    print("%-28s k1 %.4f k2 %.4f k3 %.4f  step %.4f ms  score %s  %s" % (opts, ks[0], ks[1], ks[2], step, float(s).hex(), eng.describe()[eng.describe().index("k2:"):][:90]))
    eng.close()

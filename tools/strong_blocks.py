#!/usr/bin/env python
"""Strong scaling sized on ONE GPU (round 6): every block of the 500k-family table of BASELINE configs[3]
evaluated alone (what a rank of an 8-GPU job does per evaluation) beside the whole table on one GPU -- ms per evaluation,
unique rows, states of the block's own compression plan, matrix-instruction flops issued, walk workgroups / dispatch rounds --
for the file-order deal and for rows sorted by the state of the largest compressed subtree (equal rows, and cuts balanced by
the plan model of tools/strong_blocks_plan.py).  predicted N = 8 efficiency = whole-table ms / (8 x slowest block ms)."""
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def evaluate(eng_cls, tree, rows, rng, prior, nl, nm, n_cu=256):
    eng = eng_cls(0)
    tree.apply(eng)
    t0 = time.perf_counter()
    eng.set_families(rows, rng)
    setup = time.perf_counter() - t0
    for _ in range(45):
        eng.get_posterior(nl, nm, prior)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        score, fz = eng.get_posterior(nl, nm, prior)
    ms = (time.perf_counter() - t0) / n * 1e3
    walk_fl, table_fl = eng.last_issued_flops()
    d = eng.describe()
    states = int(re.search(r"states=(\d+)", d).group(1)) if "states=" in d else 0
    nf = int(re.search(r"NF=(\d+)", d).group(1))
    m = re.search(r"grid=(\d+)", d)
    grid = int(m.group(1)) if m else (len(rows) + nf - 1) // nf
    eng.close()
    return {"ms": ms, "states": states, "flops": walk_fl + table_fl, "grid": grid, "nf": nf, "score": score, "setup_s": setup, "desc": d}


def main():
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import prior as cprior, synth, tree as ctree
    from strong_blocks_plan import plan_products
    cfg = dict(synth.CONFIGS["cfg4"])
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"]))
    tree = ctree.CafeTree(newick)
    per = 62464
    blocks = []
    for b in range(8):
        path = "/tmp/strong_block_%d_%d.npy" % (per, b)
        if os.path.exists(path):
            blocks.append(np.load(path))
        else:
            blocks.append(synth.simulate_families(tree, per, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1 + b))
            np.save(path, blocks[-1])
    table = np.concatenate(blocks)
    rng = cafe_amd.init_family_size(cfg["m"])
    prior = cprior.prior_rfsize_poisson(rng.root_min, 9.0)
    nl, nm = synth.node_rates(tree, cfg)
    whole = evaluate(cafe_amd.Engine, tree, table, rng, prior, nl, nm)
    print("whole table on one GPU: %.3f ms per evaluation, %d states, %.3f GFLOP issued, %d workgroups of %d rows; plan set-up %.2f s" % (
        whole["ms"], whole["states"], whole["flops"] / 1e9, whole["grid"], whole["nf"], whole["setup_s"]), flush=True)
    print("   " + whole["desc"][whole["desc"].index("k2:"):][:300], flush=True)

    def deal(name, perm, cuts):
        t = table[perm]
        res = []
        for b in range(8):
            rows = t[cuts[b]:cuts[b + 1]]
            r = evaluate(cafe_amd.Engine, tree, rows, rng, prior, nl, nm)
            r["rows"] = len(rows)
            res.append(r)
        tot = sum(r["ms"] for r in res)
        mx = max(r["ms"] for r in res)
        print("== %s" % name)
        for b, r in enumerate(res):
            print("   block %d: %6d rows  %.3f ms  %7d states  %.3f GFLOP  %5d workgroups of %d" % (b, r["rows"], r["ms"], r["states"], r["flops"] / 1e9, r["grid"], r["nf"]))
        print("   sum of blocks %.3f ms = %.3f x whole table; slowest %.3f ms -> predicted N = 8 efficiency %.3f (sum/8 would give %.3f)" % (
            tot, tot / whole["ms"], mx, whole["ms"] / (8 * mx), whole["ms"] / tot), flush=True)
        return res
    equal = [b * per for b in range(9)]
    deal("file order, equal rows", np.arange(len(table)), equal)
    # rows sorted by the state of the largest maximal compressed subtree of the whole table's plan (first-appearance ids)
    Fu, states, wp, ncomp, maxi = plan_products(tree, table, 0.7)
    big = max(maxi, key=lambda v: maxi[v])

    def leaves(v):
        return [v // 2] if tree.left[v] < 0 else leaves(tree.left[v]) + leaves(tree.right[v])
    sub = table[:, leaves(big)]
    _, first, inv = np.unique(sub, axis=0, return_index=True, return_inverse=True)
    rank_of = np.empty(len(first), np.int64)
    rank_of[np.argsort(first, kind="stable")] = np.arange(len(first))    # state ids in order of first appearance (deterministic)
    perm = np.argsort(rank_of[inv.ravel()], kind="stable")
    deal("sorted by the largest compressed subtree (node %d, %d states), equal rows" % (big, maxi[big]), perm, equal)
    # cuts balanced by the plan model: products of a block = states + rows x walk products
    t = table[perm]
    cuts = list(equal)
    for it in range(3):
        w = []
        for b in range(8):
            Fu_b, st_b, wp_b, _, _ = plan_products(tree, t[cuts[b]:cuts[b + 1]], 0.7)
            w.append(st_b + Fu_b * wp_b)
        dens = [w[b] / max(cuts[b + 1] - cuts[b], 1) for b in range(8)]     # work per row inside each block
        target = sum(w) / 8.0
        new = [0]
        acc_rows = 0.0
        b = 0
        pos = 0
        # walk along the rows accumulating modelled work until each target is met
        cum = np.concatenate([[0.0], np.cumsum(np.repeat(dens, np.diff(cuts)))])
        for k in range(1, 8):
            pos = int(np.searchsorted(cum, k * target))
            new.append((pos // 256) * 256)
        new.append(len(table))
        cuts = new
    deal("sorted by the largest compressed subtree, chunk-aligned cuts balanced by the plan model", perm, cuts)


if __name__ == "__main__":
    main()

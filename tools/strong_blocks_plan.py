#!/usr/bin/env python
"""How the rows of the 500k-family table (BASELINE configs[3]) are dealt to 8 blocks decides how much subtree state a block
shares (round 6).  CPU-only model of the compression plan (cafe_amd/csrc/compression_plan.hpp: a node is
compressed when both children are leaves or compressed and it has at most theta * Fu states): products per evaluation =
sum of states over compressed nodes + Fu x (walk steps with an internal child), for the whole table and for each block of
(a) the file order, (b) rows sorted lexicographically by the counts in leaf order, (c) rows sorted by the states of the
compressed forest's maximal nodes (largest subtrees first).

    python tools/strong_blocks_plan.py [F_total] [theta]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def plan_products(tree, rows, theta):
    """-> (unique rows, states over compressed nodes, walk products per row, compressed nodes, sid of maximal compressed nodes)"""
    n = tree.n_nodes
    u = np.unique(rows, axis=0)
    Fu = len(u)
    limit = int(theta * Fu)
    sid, comp, D = {}, {}, {}
    order = []
    st = [(tree.root, 0)]
    while st:
        v, stage = st.pop()
        if tree.left[v] < 0:
            order.append(v)
        elif stage == 0:
            st.append((v, 1))
            st.append((tree.right[v], 0))
            st.append((tree.left[v], 0))
        else:
            order.append(v)
    for v in order:
        if tree.left[v] < 0:
            sid[v] = u[:, v // 2].astype(np.int64)
            continue
        a, b = tree.left[v], tree.right[v]
        okc = all(tree.left[c] < 0 or comp.get(c) for c in (a, b))
        comp[v] = False
        if v != tree.root and okc:
            key = sid[a] * (int(sid[b].max()) + 1) + sid[b]
            uniq, inv = np.unique(key, return_inverse=True)
            if len(uniq) <= limit:
                comp[v], D[v], sid[v] = True, len(uniq), inv.astype(np.int64)
    states = sum(D.values())
    parent = {}
    for v in range(n):
        if tree.left[v] >= 0:
            parent[tree.left[v]] = v
            parent[tree.right[v]] = v
    under = set()
    for v in order[::-1]:
        p = parent.get(v)
        if p is not None and (comp.get(p) or p in under):
            under.add(v)
    walk_products = 0
    for v in range(n):
        if tree.left[v] < 0 or comp.get(v) or v in under:
            continue
        for c in (tree.left[v], tree.right[v]):
            if tree.left[c] >= 0 and not comp.get(c):
                walk_products += 1
    maximal = [v for v in D if v not in under]
    return Fu, states, walk_products, len(D), {v: D[v] for v in maximal}


def main():
    from cafe_amd import synth, tree as ctree
    F_total = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 62464
    theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
    cfg = dict(synth.CONFIGS["cfg4"])
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"]))
    tree = ctree.CafeTree(newick)
    per = F_total // 8
    t0 = time.time()
    blocks = []
    for b in range(8):
        path = "/tmp/strong_block_%d_%d.npy" % (per, b)
        if os.path.exists(path):
            blocks.append(np.load(path))
        else:
            blocks.append(synth.simulate_families(tree, per, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1 + b))
            np.save(path, blocks[-1])
    table = np.concatenate(blocks)
    print("table: %d rows x %d leaves (%.0f s)" % (table.shape[0], table.shape[1], time.time() - t0), flush=True)
    Fu, states, wp, ncomp, maxi = plan_products(tree, table, theta)
    whole = states + Fu * wp
    print("whole table: %d unique rows, %d compressed nodes, %d states, %d walk products per row -> %.3f M products" % (Fu, ncomp, states, wp, whole / 1e6), flush=True)

    def report(name, perm):
        t = table[perm]
        tot = 0
        line = []
        for b in range(8):
            Fu_b, st_b, wp_b, nc_b, _ = plan_products(tree, t[b * per:(b + 1) * per], theta)
            p = st_b + Fu_b * wp_b
            tot += p
            line.append("%d/%d/%d" % (Fu_b, st_b, wp_b))
        print("%-34s sum of blocks %.3f M products = %.3f x whole  (unique rows / states / walk products per row: %s)" % (name, tot / 1e6, tot / whole, "  ".join(line)), flush=True)
    report("file order", np.arange(len(table)))
    # leaf order = in-order leaves = columns 0..n_leaves-1
    report("lexicographic by leaf order", np.lexsort(table.T[::-1]))
    # by the states of the maximal compressed nodes of the WHOLE table's plan, largest first
    n = tree.n_nodes
    # recompute sid for the whole table rows (not only unique): reuse plan on all rows
    keys = []
    order_nodes = sorted(maxi, key=lambda v: -maxi[v])
    leaves_below = {}

    def leaves(v):
        if tree.left[v] < 0:
            return [v // 2]
        return leaves(tree.left[v]) + leaves(tree.right[v])
    for v in order_nodes:
        cols = leaves(v)
        sub = table[:, cols]
        _, inv = np.unique(sub, axis=0, return_inverse=True)
        keys.append(inv.ravel())
    report("by maximal compressed subtrees", np.lexsort(tuple(keys[::-1])))
    report("by the largest one only", np.argsort(keys[0], kind="stable"))


if __name__ == "__main__":
    main()

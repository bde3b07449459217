"""Synthetic inputs for the BASELINE.json configs.

Trees: random-join ultrametric, integer branch lengths >= 1, root-to-tip height 100.
Families: simulated like `genfamily` -- root size drawn, then per edge an inverse-CDF draw on
the parent's row of the birth-death transition matrix (cafe/cafe_tree.c:533-569) -- at
lambda_true; rows with any count > m rejected; one count == m forced so the ranges are pinned;
duplicate rows removed.  This is workload generation, not part of the measured path.
"""
import math

import numpy as np

from .tree import CafeTree


def random_ultrametric_newick(n_taxa, seed, height=100):
    rng = np.random.default_rng(seed)
    if n_taxa - 1 > height:
        raise ValueError("need n_taxa - 1 <= height distinct integer node heights")
    hs = np.sort(rng.choice(np.arange(1, height), size=n_taxa - 2, replace=False)) if n_taxa > 2 else np.array([], int)
    heights = list(hs) + [height]
    live = [("t%d" % i, 0) for i in range(n_taxa)]
    for h in heights:
        i, j = sorted(rng.choice(len(live), size=2, replace=False))
        (a, ha), (b, hb) = live[i], live[j]
        node = "(%s:%d,%s:%d)" % (a, h - ha, b, h - hb)
        live = [x for k, x in enumerate(live) if k not in (i, j)] + [(node, h)]
    return live[0][0]


def bd_matrix(t, lam, mu, M):
    """Transition matrix P[s, c] of the linear birth-death process (closed form of
    libtree/birthdeath.c:52-73 / :34-50, evaluated with exact lgamma -- generator use only)."""
    lg = np.array([math.lgamma(i + 1) for i in range(2 * M + 2)])

    def lnc(n, k):
        return lg[n] - lg[k] - lg[n - k]

    if mu < 0 or lam == mu:
        alpha = lam * t / (1 + lam * t)
        beta = alpha
    else:
        e = math.exp((lam - mu) * t)
        alpha = mu * (e - 1) / (lam * e - mu)
        beta = lam * (e - 1) / (lam * e - mu)
    coeff = 1 - alpha - beta
    P = np.zeros((M + 1, M + 1))
    P[0, 0] = 1.0
    la, lb, lc = math.log(alpha), math.log(beta), math.log(coeff)
    c = np.arange(M + 1)
    for s in range(1, M + 1):
        acc = np.zeros(M + 1)
        for j in range(0, s + 1):
            ok = c >= j
            cc = c[ok]
            t_ = lnc(s, j) + lnc(s + cc - 1 - j, s - 1) + (s - j) * la + (cc - j) * lb + j * lc
            acc[ok] += np.exp(t_)
        P[s] = np.clip(acc, 0, 1)
    return P


def simulate_families(tree, F, m, lam, mu, seed, root_cap=None):
    rng = np.random.default_rng(seed)
    M = m + max(50, m // 5)
    mats = {}
    for i in range(tree.n_nodes):
        if i == tree.root:
            continue
        key = int(tree.branchlength[i])
        if key not in mats:
            mats[key] = np.cumsum(bd_matrix(key, lam, mu, M), axis=1)
    order = []
    stack = [tree.root]
    while stack:
        v = stack.pop()
        order.append(v)
        if tree.left[v] >= 0:
            stack.append(tree.right[v])
            stack.append(tree.left[v])
    rows = np.zeros((0, tree.n_leaves), np.int32)
    seen = set()
    out = []
    cap = root_cap if root_cap is not None else max(1, int(0.6 * m))
    while len(out) < F:
        B = max(1024, 2 * (F - len(out)))
        root = 1 + rng.poisson(8, size=B)
        tail = rng.random(B) < 0.1
        root[tail] = rng.integers(1, cap + 1, size=tail.sum())
        root = np.minimum(root, m)
        sizes = np.zeros((B, tree.n_nodes), np.int64)
        sizes[:, tree.root] = root
        for v in order:
            if v == tree.root:
                continue
            cdf = mats[int(tree.branchlength[v])]
            ps = sizes[:, tree.parent[v]]
            u = rng.random(B)
            sizes[:, v] = np.minimum((cdf[ps] < u[:, None]).sum(axis=1), M)
        leaves = sizes[:, 0::2]
        ok = (leaves.max(axis=1) <= m)
        for row in leaves[ok]:
            key = row.tobytes()
            if key in seen:
                continue
            seen.add(key)
            out.append(row.astype(np.int32))
            if len(out) == F:
                break
    rows = np.stack(out)
    if rows.max() < m:  # pin the ranges: force one count == m
        r = rows[0].copy()
        r[0] = m
        if r.tobytes() not in seen:
            rows[0] = r
        else:
            rows[0, 0] = m
    return rows


CONFIGS = {
    # name: (F, n_taxa, m, has_mu, n_lambda_classes, seed)
    "cfg2": dict(F=10000, n_taxa=16, m=100, lam=0.002, mu=-1.0, seed=20260928,
                 desc="10k synthetic families, 16-taxon tree, max family size 100, single lambda"),
    "cfg3": dict(F=100000, n_taxa=32, m=200, lam=0.002, mu=0.0015, seed=20260929,
                 desc="100k synthetic families, 32-taxon tree, max size 200, lambdamu"),
    "cfg4": dict(F=500000, n_taxa=64, m=100, lam=0.002, mu=-1.0, seed=20260930,
                 desc="500k families, 64-taxon tree, per-clade lambda"),
}


def make_config(name, F=None):
    cfg = dict(CONFIGS[name])
    if F is not None:
        cfg["F"] = F
    newick = random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
    tree = CafeTree(newick)
    counts = simulate_families(tree, cfg["F"], cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    cfg["newick"] = newick
    return tree, counts, cfg

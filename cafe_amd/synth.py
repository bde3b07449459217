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
    la, lb, lc = math.log(alpha), math.log(beta), math.log(coeff)
    # acc[s, c] += term(s, c, j), j ascending -- the same additions per entry, in the same order, as a loop over s with
    # an inner loop over j (tests/golden/synth_table_hashes.json pins the tables), vectorised over (s, c)
    sv = np.arange(1, M + 1)[:, None]            # rows s = 1..M
    cv = np.arange(M + 1)[None, :]
    acc = np.zeros((M, M + 1))
    for j in range(0, M + 1):
        ok = (sv >= j) & (cv >= j)
        if not ok.any():
            break
        ss, cc = np.broadcast_to(sv, ok.shape)[ok], np.broadcast_to(cv, ok.shape)[ok]
        t_ = lnc(ss, j) + lnc(ss + cc - 1 - j, ss - 1) + (ss - j) * la + (cc - j) * lb + j * lc
        acc[ok] += np.exp(t_)
    P[0, 0] = 1.0
    P[1:] = np.clip(acc, 0, 1)
    return P


def _count_below(cdf, ps, u):
    """(cdf[ps] < u[:, None]).sum(axis=1) -- the number of entries of the parent's cumulative row strictly below the
    uniform draw -- without materialising the B x S comparison: rows of a cumulative sum are non-decreasing, so the
    count is the left insertion point, taken group by group of equal parent size.  Same integers (the tables of
    tests/golden/synth_table_hashes.json, recorded with the dense form, pin that)."""
    out = np.empty(len(ps), np.int64)
    order = np.argsort(ps, kind="stable")
    sp = ps[order]
    starts = np.flatnonzero(np.r_[True, sp[1:] != sp[:-1]])
    ends = np.r_[starts[1:], len(sp)]
    for a, b in zip(starts, ends):
        idx = order[a:b]
        out[idx] = np.searchsorted(cdf[sp[a]], u[idx], side="left")
    return out


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
            sizes[:, v] = np.minimum(_count_below(cdf, ps, u), M)
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


def root_size_distribution(m, n=1000, root_cap=None):
    """P(root = 1 + i), i < n, of the root draw in simulate_families: 1 + Poisson(8) with probability 0.9, uniform
    on [1, cap] with probability 0.1, clipped to m."""
    cap = root_cap if root_cap is not None else max(1, int(0.6 * m))
    p = np.zeros(max(n, m + 1))
    k = np.arange(len(p))
    pois = np.exp(k * math.log(8.0) - 8.0 - np.array([math.lgamma(i + 1.0) for i in k]))
    p += 0.9 * pois                      # index i <-> root 1 + i
    p[:cap] += 0.1 / cap
    p[m - 1] += p[m:].sum()              # np.minimum(root, m)
    p[m:] = 0.0
    return p[:n]


def clade_classes(tree, n_classes):
    """Lambda classes by clade (the `lambda -t` tree of cafe/cafe_shell.c:334-393): the two subtrees below
    the root are split, largest first, until there are n_classes clades; every node of a clade (its top
    node's branch included) gets that clade's class, clades numbered in left-to-right order.  Returns
    (class per node, 0-based; the root gets 0) and the lambda-tree text (classes 1-based on every node
    but the root, as the reference requires)."""
    n_leaves_below = np.zeros(tree.n_nodes, int)
    for v in range(0, tree.n_nodes, 2):
        u = v
        while u >= 0:
            n_leaves_below[u] += 1
            u = tree.parent[u]
    clades = [int(tree.left[tree.root]), int(tree.right[tree.root])]
    while len(clades) < n_classes:
        internal = [c for c in clades if tree.left[c] >= 0]
        if not internal:
            break
        big = max(internal, key=lambda c: n_leaves_below[c])
        i = clades.index(big)
        clades[i:i + 1] = [int(tree.left[big]), int(tree.right[big])]
    cls = np.zeros(tree.n_nodes, np.int32)
    tops = set(clades)

    def paint(v, c):
        cls[v] = c
        if tree.left[v] >= 0:
            paint(int(tree.left[v]), c)
            paint(int(tree.right[v]), c)

    # in-order ids grow left to right: number the clades by position; nodes above the clade tops (split
    # ancestors) inherit the class of their left-most clade
    for c, top in enumerate(sorted(clades)):
        paint(top, c)
    for v in range(tree.n_nodes):
        if v != tree.root and v not in tops and not any(_is_ancestor(tree, t, v) for t in tops):
            u = v
            while u not in tops:
                u = int(tree.left[u])
            cls[v] = cls[u]

    def text(v):
        lab = "" if v == tree.root else str(int(cls[v]) + 1)
        if tree.left[v] < 0:
            return lab
        return "(" + text(int(tree.left[v])) + "," + text(int(tree.right[v])) + ")" + lab

    return cls, text(tree.root)


def _is_ancestor(tree, a, v):
    while v >= 0:
        if v == a:
            return True
        v = int(tree.parent[v])
    return False


# the +-2 band of the reference's example/errormodel/error1.txt (SURVEY.md 8d cfg 5): Pr(observed = true + d)
ERROR_BAND = {-2: 0.0, -1: 0.0, 0: 0.94, 1: 0.05, 2: 0.01}


def banded_error_matrix(range_max, band=None):
    """errormatrix[observed][true] of a banded model file with `maxcnt = range_max` after the reference's
    reader and column fix-up (cafe/error_model.cpp:145-204, cafe/cafe_shell.c:585-622): the band on every
    column, the first and last `todiff` columns topped up at row 0 / row mfs so that they sum to 1 (the
    middle columns already do)."""
    band = band or ERROR_BAND
    mfs = range_max
    E = np.zeros((mfs + 1, mfs + 1))
    for true in range(mfs + 1):
        for d, p in band.items():
            if 0 <= true + d <= mfs:
                E[true + d, true] = p
    todiff = max(band)
    for c in range(0, min(todiff, mfs + 1)):
        E[0, c] += 1.0 - E[:, c].sum()
    for c in range(max(mfs - todiff + 1, 0), mfs + 1):
        E[mfs, c] += 1.0 - E[:, c].sum()
    return E


def write_error_model_file(path, range_max, band=None):
    """The same model as a file `errormodel -model` reads (cafe/error_model.cpp:145-204)."""
    band = band or ERROR_BAND
    ds = sorted(band)
    with open(path, "w") as f:
        f.write("maxcnt:%d\n" % range_max)
        f.write("cntdiff " + " ".join(str(d) for d in ds) + "\n")
        for j in range(range_max + 1):
            f.write(str(j) + " " + " ".join("%.2f" % band[d] for d in ds) + "\n")


def simulate_null_rows(tree, matrices, rng, trials, seed):
    """Rows of a Monte-Carlo null (get_random_probabilities, cafe/conditional_distribution.cpp:10-44): for every
    root size s in the root range, `trials` families simulated down the tree by inverse-CDF draws on the parent's
    matrix row (cafe/cafe_tree.c:533-569), each with root rows {s} and the running-minimum column limit of :29.
    numpy random numbers, vectorised over all rows: workload generation for bench.py and the property tests (the
    host driver draws the same structure in the reference's rand() order).  matrices: node -> S x S array.
    Returns (counts[R*trials, n_leaves], root_size[R*trials], col_max[R*trials])."""
    R = rng.root_max - rng.root_min + 1
    B = R * trials
    M = max(rng.max, rng.root_max)
    order, stack = [], [tree.root]
    while stack:
        v = stack.pop()
        order.append(v)
        if tree.left[v] >= 0:
            stack.append(tree.right[v])
            stack.append(tree.left[v])
    g = np.random.default_rng(seed)
    sizes = np.zeros((B, tree.n_nodes), np.int64)
    root_sizes = np.repeat(np.arange(rng.root_min, rng.root_max + 1), trials)
    sizes[:, tree.root] = root_sizes
    cdfs = {}
    for v in order:
        if v == tree.root:
            continue
        key = id(matrices[v])
        if key not in cdfs:
            cdfs[key] = np.cumsum(matrices[v], axis=1)
        ps = sizes[:, tree.parent[v]]
        u = g.random(B)
        sizes[:, v] = np.minimum((cdfs[key][ps] < u[:, None]).sum(axis=1), M)
    counts = np.ascontiguousarray(sizes[:, 0::2].astype(np.int32))
    mx = sizes[:, [v for v in range(tree.n_nodes) if v != tree.root]].max(axis=1)
    cm = np.minimum(mx + np.maximum(50, mx // 5), rng.max)
    cm = np.minimum.accumulate(cm.reshape(R, trials), axis=1).reshape(-1)
    return counts, root_sizes.astype(np.int32), cm.astype(np.int32)


CONFIGS = {
    # BASELINE.json configs[1..4]
    "cfg2": dict(F=10000, n_taxa=16, m=100, lam=0.002, mu=-1.0, seed=20260928, baseline_index=1,
                 desc="10k synthetic families, 16-taxon tree, max family size 100, single lambda"),
    "cfg3": dict(F=100000, n_taxa=32, m=200, lam=0.002, mu=0.0015, seed=20260929, baseline_index=2,
                 desc="100k synthetic families, 32-taxon tree, max size 200, lambdamu"),
    "cfg4": dict(F=500000, n_taxa=64, m=100, lam=0.002, mu=-1.0, seed=20260930, baseline_index=3, n_classes=3,
                 class_scale=(1.0, 1.25, 0.8),
                 desc="500k families sharded over 8 GPUs (62,500 per GPU), 64-taxon tree, 3 lambda classes by clade"),
    "cfg5": dict(F=100000, n_taxa=32, m=200, lam=0.002, mu=-1.0, seed=20261001, tree_seed=20260929, baseline_index=4,
                 error_model=True,
                 desc="100k families, 32-taxon tree (the configs[2] tree), max size 200, +-2 band error model on every "
                      "leaf, lambda search then Monte-Carlo null (R x 1000) + p-values"),
}


def node_rates(tree, cfg, scale=1.0, mu_scale=1.0):
    """Per-node (lambda, mu) of a config at `scale` x its true rates (cafe_shell_set_lambdas,
    cafe/cafe_shell.c:31-38): one lambda per clade class where the config has classes."""
    if cfg.get("n_classes"):
        cls, _ = clade_classes(tree, cfg["n_classes"])
        lam = np.array([cfg["lam"] * cfg["class_scale"][c] * scale for c in cls])
    else:
        lam = np.full(tree.n_nodes, cfg["lam"] * scale)
    mu = np.full(tree.n_nodes, cfg["mu"] * mu_scale if cfg["mu"] >= 0 else -1.0)
    return lam, mu


def make_config(name, F=None):
    cfg = dict(CONFIGS[name])
    if F is not None:
        cfg["F"] = F
    newick = random_ultrametric_newick(cfg["n_taxa"], cfg.get("tree_seed", cfg["seed"]))
    tree = CafeTree(newick)
    counts = simulate_families(tree, cfg["F"], cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1)
    cfg["newick"] = newick
    if cfg.get("n_classes"):
        cfg["node_class"], cfg["lambda_tree"] = clade_classes(tree, cfg["n_classes"])
    return tree, counts, cfg

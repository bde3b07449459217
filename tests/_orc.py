"""ctypes binding of oracle/liboracle.so (the CPU checker) for tests, smoke() and
bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE ONLY -- never imported by cafe_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def _build():
    # CAFE_ORACLE_NATIVE=1 (bench.py's cpu_baseline leg): the same source built ON THIS BOX with
    # -O3 -march=native; the tests use the portable -O2 build that travels with the repo.
    # Both builds keep -ffp-contract=off, and gcc does not reassociate without -ffast-math: same bits.
    native = os.environ.get("CAFE_ORACLE_NATIVE") == "1"
    name = "liboracle_native.so" if native else "liboracle.so"
    so = os.path.join(ORACLE_DIR, name)
    src = os.path.join(ORACLE_DIR, "cafe_oracle.c")
    if native or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-B" if native else "-s", "-C", ORACLE_DIR, name], stdout=subprocess.DEVNULL)
    return so


class Range(C.Structure):
    _fields_ = [("min", C.c_int), ("max", C.c_int), ("root_min", C.c_int), ("root_max", C.c_int)]


class Tree(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int),
        ("parent", C.POINTER(C.c_int)),
        ("left", C.POINTER(C.c_int)),
        ("right", C.POINTER(C.c_int)),
        ("branchlength", C.POINTER(C.c_double)),
        ("root", C.c_int),
    ]


MATH_FUNC = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_void_p)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(_build())
    L.orc_gammaln.restype = C.c_double
    L.orc_gammaln.argtypes = [C.c_double]
    L.orc_chooseln.restype = C.c_double
    L.orc_chooseln.argtypes = [C.c_double, C.c_double]
    L.orc_poisspdf.restype = C.c_double
    L.orc_poisspdf.argtypes = [C.c_int, C.c_double]
    L.orc_pvalue.restype = C.c_double
    L.orc_pvalue.argtypes = [C.c_double, _dp, C.c_int]
    L.orc_chooseln_table.restype = C.POINTER(C.c_double)
    L.orc_chooseln_table.argtypes = [C.c_int]
    L.orc_birthdeath_rate_with_log_alpha.restype = C.c_double
    L.orc_birthdeath_rate_with_log_alpha.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp, C.c_int]
    L.orc_birthdeath_rate_with_log_alpha_beta.restype = C.c_double
    L.orc_birthdeath_rate_with_log_alpha_beta.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, C.c_int]
    L.orc_birthdeath_likelihood_with_s_c.restype = C.c_double
    L.orc_birthdeath_likelihood_with_s_c.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, C.c_int]
    L.orc_compute_birthdeath_rates.restype = None
    L.orc_compute_birthdeath_rates.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, _dp]
    L.orc_square_matrix_multiply.restype = None
    L.orc_square_matrix_multiply.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    L.orc_matrices_build.restype = C.c_void_p
    L.orc_matrices_build.argtypes = [C.POINTER(Tree), _dp, _dp, C.c_int, C.c_int]
    L.orc_matrices_nkeys.restype = C.c_int
    L.orc_matrices_nkeys.argtypes = [C.c_void_p]
    L.orc_matrices_size.restype = C.c_int
    L.orc_matrices_size.argtypes = [C.c_void_p]
    L.orc_matrices_get.restype = C.POINTER(C.c_double)
    L.orc_matrices_get.argtypes = [C.c_void_p, C.c_int]
    L.orc_matrices_free.restype = None
    L.orc_matrices_free.argtypes = [C.c_void_p]
    L.orc_compute_tree_likelihoods.restype = None
    L.orc_compute_tree_likelihoods.argtypes = [C.POINTER(Tree), C.POINTER(Range), C.c_void_p, _ip, _dp, C.c_int,
                                               C.POINTER(C.c_ubyte), _dp, C.c_int]
    L.orc_compute_posterior.restype = None
    L.orc_compute_posterior.argtypes = [_dp, C.c_int, _dp, _dp, _ip, _dp]
    L.orc_eval_posterior.restype = C.c_double
    L.orc_eval_posterior.argtypes = [C.POINTER(Tree), C.c_int, C.c_int, _ip, _ip, C.POINTER(Range), _dp, _dp, _dp,
                                     _dp, C.c_int, C.POINTER(C.c_ubyte), C.c_int, _ip, _dp, _ip, _dp]
    L.orc_eval_root_likelihoods.restype = None
    L.orc_eval_root_likelihoods.argtypes = [C.POINTER(Tree), C.c_int, C.c_int, _ip, _ip, _ip, _ip, C.c_void_p, _dp]
    L.orc_init_family_size.restype = None
    L.orc_init_family_size.argtypes = [C.POINTER(Range), C.c_int]
    L.orc_family_check_the_pattern.restype = None
    L.orc_family_check_the_pattern.argtypes = [C.c_int, C.c_int, _ip, _ip]
    L.orc_prior_poisson.restype = None
    L.orc_prior_poisson.argtypes = [_dp, C.c_int, C.c_int, C.c_double]
    L.orc_lnLPoisson.restype = C.c_double
    L.orc_lnLPoisson.argtypes = [C.c_double, C.c_int, C.c_int, _ip]
    L.orc_find_poisson_lambda.restype = C.c_double
    L.orc_find_poisson_lambda.argtypes = [C.c_int, C.c_int, _ip, C.c_double, _ip, _dp]
    L.orc_fminsearch.restype = C.c_int
    L.orc_fminsearch.argtypes = [MATH_FUNC, C.c_int, C.c_void_p, _dp, C.c_double, C.c_double, C.c_int, _dp, _dp, _ip]
    L.orc_conditional_distribution.restype = None
    L.orc_conditional_distribution.argtypes = [C.POINTER(Tree), C.POINTER(Range), C.c_void_p, C.c_int, _dp]
    L.orc_tree_random_familysize.restype = C.c_int
    L.orc_tree_random_familysize.argtypes = [C.POINTER(Tree), C.c_void_p, C.c_int, C.c_int, _ip]
    L.orc_tree_viterbi.restype = None
    L.orc_tree_viterbi.argtypes = [C.POINTER(Tree), C.POINTER(Range), C.c_void_p, _ip, _ip, _dp, C.c_int]
    L.orc_tree_p_values.restype = None
    L.orc_tree_p_values.argtypes = [C.POINTER(Tree), C.POINTER(Range), C.c_void_p, _ip, _dp, C.c_int, _dp]
    L.orc_viterbi_sum_probabilities.restype = None
    L.orc_viterbi_sum_probabilities.argtypes = [C.POINTER(Tree), C.POINTER(Range), C.c_void_p, _ip, _dp]
    L.orc_family_forced_range.restype = None
    L.orc_family_forced_range.argtypes = [C.POINTER(Range), C.c_int, _ip]
    _lib = L
    return L


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


# ---------------------------------------------------------------------------
# A tiny independent Newick reader for the tests (topology + branch lengths in the
# reference's in-order nlist numbering, cafe/cafe_commands.cpp:2028-2051).
# ---------------------------------------------------------------------------
class PyTree:
    def __init__(self, newick):
        s = newick.strip().rstrip(";")
        self.names, self.bl, self.children = [], [], []
        pos = [0]

        def parse():
            node = len(self.names)
            self.names.append("")
            self.bl.append(-1.0)
            self.children.append([])
            if s[pos[0]] == "(":
                pos[0] += 1
                while True:
                    ch = parse()
                    self.children[node].append(ch)
                    if s[pos[0]] == ",":
                        pos[0] += 1
                        continue
                    assert s[pos[0]] == ")"
                    pos[0] += 1
                    break
            j = pos[0]
            while j < len(s) and s[j] not in ",():":
                j += 1
            self.names[node] = s[pos[0]:j]
            pos[0] = j
            if j < len(s) and s[j] == ":":
                k = j + 1
                while k < len(s) and s[k] not in ",()":
                    k += 1
                self.bl[node] = float(s[j + 1:k])
                pos[0] = k
            return node

        root = parse()
        order = []

        def inorder(n):
            if self.children[n]:
                assert len(self.children[n]) == 2, "binary trees only"
                inorder(self.children[n][0])
                order.append(n)
                inorder(self.children[n][1])
            else:
                order.append(n)

        inorder(root)
        idmap = {old: new for new, old in enumerate(order)}
        n = len(order)
        self.n_nodes = n
        self.parent = np.full(n, -1, np.int32)
        self.left = np.full(n, -1, np.int32)
        self.right = np.full(n, -1, np.int32)
        self.branchlength = np.full(n, -1.0, np.float64)
        self.name = [""] * n
        for old in range(n):
            i = idmap[old]
            self.name[i] = self.names[old]
            self.branchlength[i] = self.bl[old]
            if self.children[old]:
                a, b = self.children[old]
                self.left[i], self.right[i] = idmap[a], idmap[b]
                self.parent[idmap[a]] = i
                self.parent[idmap[b]] = i
        self.root = idmap[root]
        self.leaf_names = [self.name[i] for i in range(0, n, 2)]
        self.n_leaves = (n + 1) // 2

    def ctree(self):
        t = Tree()
        t.n_nodes = self.n_nodes
        t.parent = iptr(self.parent)
        t.left = iptr(self.left)
        t.right = iptr(self.right)
        t.branchlength = dptr(self.branchlength)
        t.root = self.root
        self._keep = t
        return t


def make_range(mn, mx, rmin, rmax):
    r = Range()
    r.min, r.max, r.root_min, r.root_max = mn, mx, rmin, rmax
    return r


def range_from_max(m):
    r = Range()
    lib().orc_init_family_size(C.byref(r), int(m))
    return r


def eval_posterior(tree, counts, rng, node_lambda, node_mu, prior, ref=None, errormatrix=None, err_mfs=0,
                   leaf_has_err=None, nthreads=1):
    L = lib()
    counts = np.ascontiguousarray(counts, np.int32)
    F, nl = counts.shape
    nlam = np.ascontiguousarray(node_lambda, np.float64)
    nmu = np.ascontiguousarray(node_mu, np.float64)
    prior = np.ascontiguousarray(prior, np.float64)
    ml = np.zeros(F)
    mp = np.zeros(F)
    am = np.zeros(F, np.int32)
    fz = C.c_int(-1)
    refp = iptr(np.ascontiguousarray(ref, np.int32)) if ref is not None else None
    errp = dptr(np.ascontiguousarray(errormatrix, np.float64)) if errormatrix is not None else None
    lhe = None
    if leaf_has_err is not None:
        lhe_arr = np.ascontiguousarray(leaf_has_err, np.uint8)
        lhe = lhe_arr.ctypes.data_as(C.POINTER(C.c_ubyte))
    t = tree.ctree()
    score = L.orc_eval_posterior(C.byref(t), F, nl, iptr(counts), refp, C.byref(rng), dptr(nlam), dptr(nmu),
                                 dptr(prior), errp, err_mfs, lhe, nthreads, C.byref(fz), dptr(ml), iptr(am), dptr(mp))
    return score, fz.value, ml, am, mp


def birthdeath_matrix(bl, lam, mu, M):
    out = np.zeros((M + 1, M + 1))
    lib().orc_compute_birthdeath_rates(float(bl), float(lam), float(mu), int(M), dptr(out))
    return out


def prior_poisson(n, shift, lam):
    p = np.zeros(n)
    lib().orc_prior_poisson(dptr(p), n, shift, float(lam))
    return p


def load_family_table(path, max_size=-1, sep="\t"):
    """Family table reader following cafe/gene_family.cpp:186-225: header 'Desc ID sp...', rows
    kept when max(count) <= max_size (or max_size < 0)."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        header = f.readline().rstrip("\r\n").split(sep)
        species = header[2:]
        ids, rows = [], []
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            parts = line.split(sep)
            vals = [int(x) for x in parts[2:]]
            if max_size < 0 or max(vals) <= max_size:
                ids.append(parts[1])
                rows.append(vals)
    return species, ids, np.asarray(rows, np.int32)


def reorder_to_tree(species, counts, tree):
    """Column for leaf slot j <-> node 2j; species matched case-insensitively by name
    (cafe/gene_family.cpp:413-445)."""
    low = [s.lower() for s in species]
    cols = [low.index(nm.lower()) for nm in tree.leaf_names]
    return np.ascontiguousarray(counts[:, cols])


def parse_cafe_report(path):
    """Family rows of a `.cafe` text report (cafe/reports.cpp:357-388): id -> (sizes by node in
    Newick order, family-wide p, [(p1, p2) or None per internal node])."""
    import re
    out = {}
    for line in open(path):
        parts = line.rstrip("\n").split("\t")
        if len(parts) < 4 or not parts[1].startswith("(") or "_" not in parts[1]:
            continue
        sizes = [int(x) for x in re.findall(r"_(\d+)", parts[1])]
        pairs = []
        for a, b in re.findall(r"\(([^(),]+),([^(),]+)\)", parts[3]):
            pairs.append(None if a == "-" else (float(a), float(b)))
        out[parts[0]] = (sizes, float(parts[2]), pairs)
    return out


def report_with_oracle(tree, counts, rng, lam_value, trials=1000, pvalue_cut=0.05, rng_skip=2, seed=10):
    """The report pipeline on the CPU oracle: MC null with libc rand() (seed, then `rng_skip` draws as the
    Poisson-fit start and the Nelder-Mead start consume them), per-family p-values, Viterbi sizes and
    branch p-values.  Returns per family (max_p, node sizes[n_nodes], branch p[n_nodes-1] or None)."""
    L = lib()
    libc = C.CDLL(None)
    lam = np.full(tree.n_nodes, lam_value)
    mu = np.full(tree.n_nodes, -1.0)
    ct = tree.ctree()
    M = max(rng.max, rng.root_max)
    h = L.orc_matrices_build(C.byref(ct), dptr(lam), dptr(mu), M, 1)
    libc.srand(seed)
    for _ in range(rng_skip):
        libc.rand()
    R = rng.root_max - rng.root_min + 1
    cd = np.zeros((R, trials))
    L.orc_conditional_distribution(C.byref(ct), C.byref(rng), h, trials, dptr(cd))
    sof = M + 2
    vit = np.zeros(tree.n_nodes * sof, np.int32)
    Lb = np.zeros(tree.n_nodes * sof)
    out = []
    for i in range(counts.shape[0]):
        r = Range()
        row = np.ascontiguousarray(counts[i], np.int32)
        L.orc_family_forced_range(C.byref(r), counts.shape[1], iptr(row))
        fs = np.full(tree.n_nodes, -1, np.int32)
        fs[0::2] = row
        pv = np.zeros(max(r.root_max, 1))
        L.orc_tree_p_values(C.byref(ct), C.byref(r), h, iptr(fs), dptr(cd), trials, dptr(pv))
        maxp = float(pv[:max(r.root_max, 0)].max()) if r.root_max >= 1 else 0.0
        L.orc_tree_viterbi(C.byref(ct), C.byref(r), h, iptr(fs), iptr(vit), dptr(Lb), sof)
        bp = None
        if not (maxp > pvalue_cut):
            bp = np.zeros(tree.n_nodes - 1)
            L.orc_viterbi_sum_probabilities(C.byref(ct), C.byref(r), h, iptr(fs), dptr(bp))
        out.append((maxp, fs.copy(), bp))
    L.orc_matrices_free(h)
    return out, cd


def load_error_model(path, range_max):
    """Error-model file -> errormatrix[(mfs+1)^2] as the reference builds it: reader
    cafe/error_model.cpp:145-204 (band on the diagonal, missing tail rows copy the previous band),
    then __check_error_model_columnsums cafe/cafe_shell.c:585-622 -- including its use of the
    INTEGER abs() on a double, which leaves middle columns untouched unless |1 - sum| >= 1."""
    lines = [l.rstrip("\r\n") for l in open(path)]
    mfs = max(range_max, int(lines[0].split(" ")[0].split(":")[1]))
    head = lines[1].split(" ")
    fromdiff, todiff = int(head[1]), int(head[-1])
    E = np.zeros((mfs + 1, mfs + 1))
    j = 0
    for line in lines[2:]:
        data = line.split(" ")
        if len(data) != (todiff - fromdiff) + 2:
            continue
        assert int(data[0]) == j
        for k, i in enumerate(range(fromdiff, todiff + 1), start=1):
            if 0 <= i + j <= mfs:
                E[i + j, j] = float(data[k])
        j += 1
    while j and j <= mfs:
        for i in range(fromdiff, todiff + 1):
            if 0 <= i + j <= mfs:
                E[i + j, j] = E[i + j - 1, j - 1]
        j += 1
    diff = todiff

    def colsum(c):
        s = 0.0
        for i in range(mfs + 1):
            s += E[i, c]
        return s

    for c in range(0, min(diff, mfs + 1)):
        E[0, c] += 1 - colsum(c)
    for c in range(diff, mfs - diff + 1):
        s = colsum(c)
        if abs(int(1 - s)) > 1e-14:
            E[:, c] /= s
    for c in range(max(mfs - diff + 1, 0), mfs + 1):
        E[mfs, c] += 1 - colsum(c)
    return E, mfs


def mc_null_rows(tree, rng, mats, trials, seed, root_sizes):
    """The simulated families of the Monte-Carlo null for SOME root sizes, drawn in the reference's `-t 1`
    order (root sizes ascending, trials sequential, libc rand(): cafe/conditional_distribution.cpp:10-44,
    cafe/cafe_tree.c:533-569) -- the whole stream is consumed so that the selected root sizes see the draws
    they would see in orc_conditional_distribution.  Returns {s: (counts[trials, n_leaves], col_max[trials])}."""
    L = lib()
    libc = C.CDLL(None)
    libc.srand(seed)
    ct = tree.ctree()
    fs = np.zeros(tree.n_nodes, np.int32)
    want = set(int(s) for s in root_sizes)
    out = {}
    n_leaves = (tree.n_nodes + 1) // 2
    for s in range(rng.root_min, rng.root_max + 1):
        mfs = max(s, rng.max)
        rmax = rng.max
        keep = s in want
        if keep:
            cnt = np.zeros((trials, n_leaves), np.int32)
            cm = np.zeros(trials, np.int32)
        for i in range(trials):
            mx = L.orc_tree_random_familysize(C.byref(ct), mats, s, mfs, iptr(fs))
            rmax = min(mx + max(50, mx // 5), rmax)
            if keep:
                cnt[i] = fs[0::2]
                cm[i] = rmax
        if keep:
            out[s] = (cnt, cm)
    return out


def eval_root_likelihoods(tree, mats, counts, root_lo, root_hi, col_max, nthreads=1):
    """orc_eval_root_likelihoods over row blocks on worker threads (ctypes releases the GIL; rows are independent)."""
    from concurrent.futures import ThreadPoolExecutor
    L = lib()
    counts = np.ascontiguousarray(counts, np.int32)
    lo = np.ascontiguousarray(root_lo, np.int32)
    hi = np.ascontiguousarray(root_hi, np.int32)
    cm = np.ascontiguousarray(col_max, np.int32)
    B, nl = counts.shape
    off = np.concatenate([[0], np.cumsum(hi - lo + 1)])
    out = np.zeros(int(off[-1]))
    ct = tree.ctree()
    edges = np.linspace(0, B, max(1, min(nthreads, B)) + 1).astype(int)

    def work(k):
        a, b = int(edges[k]), int(edges[k + 1])
        if b > a:
            L.orc_eval_root_likelihoods(C.byref(ct), b - a, nl, iptr(counts[a:b]), iptr(lo[a:b]), iptr(hi[a:b]),
                                        iptr(cm[a:b]), mats, dptr(out[off[a]:off[b]]))

    with ThreadPoolExecutor(len(edges) - 1) as ex:
        list(ex.map(work, range(len(edges) - 1)))
    return out


def build_matrices(tree, rng, node_lambda, node_mu, nthreads=1):
    """orc_matrices handle for (node_lambda, node_mu) at M = max(range.max, range.root_max); free with free_matrices."""
    ct = tree.ctree()
    M = max(rng.max, rng.root_max)
    return lib().orc_matrices_build(C.byref(ct), dptr(np.ascontiguousarray(node_lambda, np.float64)),
                                    dptr(np.ascontiguousarray(node_mu, np.float64)), M, nthreads)


def free_matrices(h):
    lib().orc_matrices_free(h)


def viterbi_and_pvalues(tree, mats, row, cd, trials, pvalue_cut):
    """One family of the report on the oracle: (max p over root sizes, Viterbi node sizes, branch p-values or None)."""
    L = lib()
    ct = tree.ctree()
    r = Range()
    row = np.ascontiguousarray(row, np.int32)
    L.orc_family_forced_range(C.byref(r), len(row), iptr(row))
    fs = np.full(tree.n_nodes, -1, np.int32)
    fs[0::2] = row
    pv = np.zeros(max(r.root_max, 1))
    L.orc_tree_p_values(C.byref(ct), C.byref(r), mats, iptr(fs), dptr(cd), trials, dptr(pv))
    maxp = float(pv[:max(r.root_max, 0)].max()) if r.root_max >= 1 else 0.0
    sof = L.orc_matrices_size(mats) + 1
    vit = np.zeros(tree.n_nodes * sof, np.int32)
    Lb = np.zeros(tree.n_nodes * sof)
    L.orc_tree_viterbi(C.byref(ct), C.byref(r), mats, iptr(fs), iptr(vit), dptr(Lb), sof)
    bp = None
    if not (maxp > pvalue_cut):
        bp = np.zeros(tree.n_nodes - 1)
        L.orc_viterbi_sum_probabilities(C.byref(ct), C.byref(r), mats, iptr(fs), dptr(bp))
    return maxp, fs, bp


def clustered_posterior(tree, counts, rng, node_lambdas, node_mus, weights, prior, ref=None, nthreads=1):
    """cafe_get_clustered_posterior on the oracle: rates [K, n_nodes], weights [K].
    Returns (score, first_zero, MAP[F], p_z[F, K], new_weights[K])."""
    L = lib()
    L.orc_eval_clustered_posterior.restype = C.c_double
    L.orc_eval_clustered_posterior.argtypes = [C.POINTER(Tree), C.c_int, C.c_int, _ip, _ip, C.POINTER(Range), C.c_int, _dp, _dp,
                                               _dp, _dp, C.c_int, _dp, _dp, _dp, _ip]
    counts = np.ascontiguousarray(counts, np.int32)
    F, nl = counts.shape
    lam = np.ascontiguousarray(node_lambdas, np.float64)
    mu = np.ascontiguousarray(node_mus, np.float64)
    w = np.ascontiguousarray(weights, np.float64)
    K = lam.shape[0]
    pr = np.ascontiguousarray(prior, np.float64)
    MAP = np.zeros(F)
    pz = np.zeros((F, K))
    neww = np.zeros(K)
    fz = C.c_int(-1)
    refp = iptr(np.ascontiguousarray(ref, np.int32)) if ref is not None else None
    ct = tree.ctree()
    score = L.orc_eval_clustered_posterior(C.byref(ct), F, nl, iptr(counts), refp, C.byref(rng), K, dptr(lam), dptr(mu), dptr(w),
                                           dptr(pr), nthreads, dptr(MAP), dptr(pz), dptr(neww), C.byref(fz))
    return score, fz.value, MAP, pz, neww


def copy_weights(parameters, start, count):
    L = lib()
    L.orc_copy_weights.restype = None
    L.orc_copy_weights.argtypes = [_dp, _dp, C.c_int, C.c_int]
    p = np.ascontiguousarray(parameters, np.float64)
    out = np.zeros(count)
    L.orc_copy_weights(dptr(out), dptr(p), start, count)
    return out

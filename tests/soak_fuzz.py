#!/usr/bin/env python
"""Randomised soak of the C ABI against the oracle (test infrastructure, kept under tests/ because it uses oracle/; not
collected by pytest: run it by hand on a GPU box).
Every iteration draws a tree (3..72 taxa; balanced / caterpillar / random joins), ranges, a table (duplicates, zero rows,
conserved rows on large trees so that the reference's unscaled likelihood stays a normal double), a rate model (one
lambda, lambda/mu, per-node), sometimes a banded error model on all or some species, and kernel options, then checks
  * per-family max likelihood / max posterior / argmax and the first-zero index against the oracle (1e-9 relative),
  * compressed walk == uncompressed walk, several parameter sets in one pass == single evaluations (bit for bit),
  * batch mode (per-row root range and column limit): trimmed == untrimmed bit for bit, and the oracle (1e-9);
  * the k-cluster model (2-4 clusters): per-family MAP, memberships, score and new weights against the oracle (1e-9);
  * Viterbi node sizes (K4) of a sample of the batch rows against the oracle (reported, a last-bit tie may differ).
Usage: python tests/soak_fuzz.py [seconds] [first_seed]     -> one line per iteration, a summary, exit code 1 on a mismatch"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _orc as O  # noqa: E402
from tests.test_gpu_fuzz import random_newick  # noqa: E402


STATS = {"viterbi_rows": 0, "viterbi_differ": 0, "viterbi_examples": [], "cluster_checks": 0}


def one(seed):
    import cafe_amd
    from cafe_amd import synth
    rs = np.random.RandomState(seed)
    n = int(rs.choice([3, 4, 5, 7, 8, 12, 16, 21, 32, 40, 64, 72]))
    shape = str(rs.choice(["balanced", "caterpillar", "random"]))
    t = O.PyTree(random_newick(rs, n, shape))
    mx = int(rs.choice([5, 9, 17, 30, 41, 63, 90, 129, 150]))
    rmin = int(rs.choice([1, 1, 1, 2]))
    rmax = int(max(rmin + 1, rs.randint(2, int(1.25 * mx) + 2)))
    F = int(np.exp(rs.uniform(0, np.log(3000))))
    top = max(1, min(mx - 1, 25))
    big = n > 40
    if big or rs.rand() < 0.3:
        size = rs.randint(0, top + 1, size=(F, 1))
        counts = (size + (rs.rand(F, n) < 0.05) * rs.choice([-1, 1], size=(F, n))).clip(0, top)
        base = (0.02 if big else 0.2) / max(t.branchlength.max(), 1)
    else:
        counts = rs.poisson(rs.uniform(0.5, 4), size=(F, n)).clip(0, top)
        base = 0.4 / max(t.branchlength.max(), 1)
    counts = counts.astype(np.int32)
    if F > 3 and rs.rand() < 0.5:
        counts[rs.randint(0, F, size=F // 3)] = counts[rs.randint(0, F, size=F // 3)]   # duplicates
    if rs.rand() < 0.3:
        counts[rs.randint(0, F)] = 0
    model = str(rs.choice(["lambda", "lambdamu", "pernode"]))
    if model == "lambda":
        lam, mu = np.full(t.n_nodes, base), np.full(t.n_nodes, -1.0)
    elif model == "lambdamu":
        lam, mu = np.full(t.n_nodes, base), np.full(t.n_nodes, base * rs.uniform(0.3, 1.2))
    else:
        lam = base * (0.5 + rs.rand(t.n_nodes))
        mu = np.full(t.n_nodes, -1.0) if rs.rand() < 0.5 else base * (0.3 + rs.rand(t.n_nodes))
    rng = O.make_range(0, mx, rmin, rmax)
    prior = O.prior_poisson(1000, max(rmin, 1), float(rs.uniform(1.0, 8.0)))
    err = leaf_err = None
    if rs.rand() < 0.3:
        err = synth.banded_error_matrix(mx)
        leaf_err = np.ones(n, np.uint8) if rs.rand() < 0.5 else (rs.rand(n) < 0.5).astype(np.uint8)
        node_err = np.zeros(t.n_nodes, np.uint8)     # by node id (leaves are the even ids), as the ABI and the oracle take it
        node_err[0::2] = leaf_err
    opts = {"mfma": str(rs.choice(["auto", "4", "16"])), "errfold": str(rs.randint(0, 2)), "errband": str(rs.randint(0, 2)),
            "k2c_batch": str(rs.randint(0, 2)), "k2slots": str(rs.randint(0, 2))}
    tag = "seed %d n=%d %s mx=%d R=[%d,%d] F=%d %s err=%s %s" % (seed, n, shape, mx, rmin, rmax, F, model,
                                                              "no" if err is None else ("all" if leaf_err.all() else "some"),
                                                              " ".join("%s=%s" % kv for kv in sorted(opts.items())))
    fr = cafe_amd.FamilySizeRange(0, mx, rmin, rmax)
    res = {}
    for comp in ("1", "0"):
        eng = cafe_amd.Engine(0)
        try:
            for k, v in opts.items():
                if not (k == "mfma" and v == "auto"):
                    eng.set_option(k, v)
            eng.set_option("compress", comp)
            eng.set_tree(t.parent, t.left, t.right, t.branchlength)
            eng.set_families(counts, fr)
            if err is not None:
                eng.set_error_model(err, node_err)
            res[comp] = eng.get_posterior(lam, mu, prior, per_family=True)
            if comp == "1":
                nl = np.array([lam * f for f in (1.0, 0.7, 1.4)])
                nm = np.array([np.where(mu < 0, -1.0, mu * f) for f in (1.0, 0.7, 1.4)])
                singles = [eng.get_posterior(nl[i], nm[i], prior) for i in range(3)]
                ms, mz = eng.get_posterior_multi(nl, nm, prior)
                for i in range(3):
                    s1, z1 = singles[i]
                    assert (ms[i] == s1 or (np.isinf(ms[i]) and np.isinf(s1))) and mz[i] == z1, "multi-set differs: " + tag
            if comp == "1" and err is None and rs.rand() < 0.5:
                # k-cluster model (cafe_get_clustered_posterior): K rate sets with weights, against the oracle
                K = int(rs.randint(2, 5))
                kl = np.array([lam * f for f in rs.uniform(0.5, 1.6, size=K)])
                km = np.array([np.where(mu < 0, -1.0, mu * f) for f in rs.uniform(0.5, 1.6, size=K)])
                w = rs.rand(K) + 0.1
                w /= w.sum()
                so_k, fz_k, MAPo, pzo, newo = O.clustered_posterior(t, counts, rng, kl, km, w, prior, nthreads=os.cpu_count() or 1)
                s_k, fzg_k, memb, MAPg, pzg = eng.clustered_posterior(kl, km, w, prior, per_family=True)
                assert fzg_k == fz_k, "cluster first zero %d vs %d: %s" % (fzg_k, fz_k, tag)
                if fz_k < 0 and np.isfinite(so_k):
                    nzk = MAPo > 0
                    assert np.max(np.abs(MAPg[nzk] - MAPo[nzk]) / MAPo[nzk], initial=0) < 1e-9, "cluster MAP: " + tag
                    assert np.max(np.abs(pzg - pzo), initial=0) < 1e-9, "cluster memberships: " + tag
                    assert abs(s_k - so_k) <= 1e-9 * abs(so_k), "cluster score %r vs %r: %s" % (s_k, so_k, tag)
                    assert np.allclose(memb / len(counts), newo, rtol=1e-9, atol=1e-12), "cluster weights: " + tag
                    STATS["cluster_checks"] += 1
            if comp == "1" and err is None:
                B = int(rs.randint(1, 700))
                rows = rs.randint(0, top + 1, size=(B, n)).astype(np.int32) if not big else counts[rs.randint(0, F, size=B)]
                cm = rs.randint(max(3, top), mx + 1, size=B).astype(np.int32)
                cm[: B // 2] = np.sort(cm[: B // 2])
                lo = rs.randint(rmin, rmax + 1, size=B).astype(np.int32)
                hi = np.minimum(lo + (rs.rand(B) < 0.3) * rs.randint(0, 9, size=B), rmax).astype(np.int32)
                eng.reset_birthdeath_cache(lam, mu)
                trimmed = eng.eval_root_likelihoods(rows, lo, hi, cm)
                eng.set_option("batch_trim", 0)
                plain = eng.eval_root_likelihoods(rows, lo, hi, cm)
                assert np.array_equal(trimmed, plain), "trimmed batch differs: " + tag
                mats = O.build_matrices(t, rng, lam, mu, nthreads=4)
                try:
                    ref = O.eval_root_likelihoods(t, mats, rows, lo, hi, cm, nthreads=os.cpu_count() or 1)
                finally:
                    O.free_matrices(mats)
                nz = ref > 0
                assert np.array_equal(trimmed == 0, ~nz), "batch zero pattern: " + tag
                worst_b = float(np.max(np.abs(trimmed[nz] - ref[nz]) / ref[nz], initial=0))
                assert worst_b < 1e-9, "batch vs oracle %.3g: %s" % (worst_b, tag)
                # K4: Viterbi node sizes of up to 64 of the rows under per-row ranges, against the oracle's max-product +
                # backtrack on fresh tables (a tie decided by the last bit may legitimately differ: counted, not failed)
                nv = min(B, 64)
                rmx = rows[:nv].max(axis=1)
                vlo = np.full(nv, rmin, np.int32)
                vhi = np.maximum(vlo, np.minimum(np.rint(rmx * 1.25), rmax)).astype(np.int32)
                vcm = np.minimum(rmx + np.maximum(50, rmx // 5), mx).astype(np.int32)
                got = eng.viterbi(rows[:nv], vlo, vhi, vcm)
                import ctypes as C
                Lo = O.lib()
                ct = t.ctree()
                Mm = max(rng.max, rng.root_max)
                hmat = Lo.orc_matrices_build(C.byref(ct), O.dptr(np.ascontiguousarray(lam)), O.dptr(np.ascontiguousarray(mu)), Mm, 1)
                sof = Mm + 2
                vit = np.zeros(t.n_nodes * sof, np.int32)
                Lb = np.zeros(t.n_nodes * sof)
                for i in range(nv):
                    r = O.make_range(0, int(vcm[i]), int(vlo[i]), int(vhi[i]))
                    fs = np.full(t.n_nodes, -1, np.int32)
                    fs[0::2] = rows[i]
                    vit[:] = 0
                    Lo.orc_tree_viterbi(C.byref(ct), C.byref(r), hmat, O.iptr(fs), O.iptr(vit), O.dptr(Lb), sof)
                    STATS["viterbi_rows"] += 1
                    if list(got[i]) != list(fs):
                        STATS["viterbi_differ"] += 1
                        STATS["viterbi_examples"].append("%s row %d" % (tag[:60], i))
                Lo.orc_matrices_free(hmat)
        finally:
            eng.close()
    (s1, z1, ml1, am1, mp1), (s0, z0, ml0, am0, mp0) = res["1"], res["0"]
    assert z1 == z0 and np.array_equal(ml1, ml0) and np.array_equal(mp1, mp0) and np.array_equal(am1, am0) and \
        (s1 == s0 or (np.isinf(s1) and np.isinf(s0))), "compressed walk differs: " + tag
    kw = {}
    if err is not None:
        kw = dict(errormatrix=err, err_mfs=err.shape[0] - 1, leaf_has_err=node_err)
    so, fzo, mlo, amo, mpo = O.eval_posterior(t, counts, rng, lam, mu, prior, nthreads=os.cpu_count() or 1, **kw)
    assert z1 == fzo, "first zero %d vs %d: %s" % (z1, fzo, tag)
    nz = mlo > 0
    assert np.array_equal(ml1 == 0, ~nz), "zero pattern: " + tag
    w1 = float(np.max(np.abs(ml1[nz] - mlo[nz]) / mlo[nz], initial=0))
    w2 = float(np.max(np.abs(mp1[nz] - mpo[nz]) / mpo[nz], initial=0))
    assert w1 < 1e-9 and w2 < 1e-9, "vs oracle %.3g %.3g: %s" % (w1, w2, tag)
    assert np.all((am1 == amo) | ~nz), "argmax: " + tag
    return tag, max(w1, w2)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    torch.cuda.init()
    t0, n, worst, bad = time.time(), 0, 0.0, 0
    while time.time() - t0 < seconds:
        try:
            tag, w = one(seed)
            worst = max(worst, w)
            print("ok   %.2e  %s" % (w, tag), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL %s" % e, flush=True)
        n += 1
        seed += 1
    print("soak: %d iterations, %d failures, worst relative error vs the oracle %.3g, %.0f s" % (n, bad, worst, time.time() - t0))
    print("soak: k-cluster evaluations checked against the oracle: %d" % STATS["cluster_checks"])
    print("soak: Viterbi node sizes identical to the oracle's on %d of %d rows%s"
          % (STATS["viterbi_rows"] - STATS["viterbi_differ"], STATS["viterbi_rows"],
             "" if not STATS["viterbi_differ"] else "; differing: " + "; ".join(STATS["viterbi_examples"][:8])))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

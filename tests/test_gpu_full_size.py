"""BASELINE.json full sizes -- configs[1] (10 k families, 16 taxa), configs[2] (100 k families, 32 taxa,
lambda/mu), one GPU shard of configs[3] (500 k / 8 = 62,500 families, 64 taxa, three lambda classes by clade) and
configs[4] (100 k families, error model on every leaf, Monte-Carlo null 250 x 1000, report) -- through
size-independent properties, plus oracle spot checks on random samples:
  * a family's values do not depend on which batch it is evaluated in (bit-exact);
  * an identity error model is the same as no error model (the one-hot leaf GEMM adds exact zeros);
  * chunk-aligned shards evaluated separately combine to the single-table score bit for bit;
  * sum of the per-family log posteriors equals the reported score.
"""
import math
import os

import numpy as np
import pytest

from tests import _orc as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[("cfg2", 10000), ("cfg3", 100000), ("cfg4", 62500), ("cfg5", 100000)],
                ids=["configs1-10k", "configs2-100k", "configs3-shard-62k-3-lambda-classes", "configs4-100k-errormodel"])
def problem(request):
    import torch
    torch.cuda.init()  # torch's bundled HIP runtime must come up before libcafehip's (as in bench.py)
    import cafe_amd
    from cafe_amd import synth
    name, F = request.param
    tree, counts, cfg = synth.make_config(name, F=F)
    rng = O.range_from_max(cfg["m"])
    t = O.PyTree(cfg["newick"])
    prior = O.prior_poisson(1000, rng.root_min, 8.0)
    assert np.array_equal(t.parent, tree.parent) and np.array_equal(t.left, tree.left)   # same nlist numbering
    lam, mu = synth.node_rates(tree, cfg)   # one lambda per clade class for configs[3]
    if cfg.get("n_classes"):
        assert len(set(lam.tolist())) == cfg["n_classes"]
    eng = cafe_amd.Engine(0)
    eng.set_tree(t.parent, t.left, t.right, t.branchlength)
    fr = cafe_amd.FamilySizeRange(rng.min, rng.max, rng.root_min, rng.root_max)
    eng.set_families(counts, fr)
    err = synth.banded_error_matrix(rng.max) if cfg.get("error_model") else None
    if err is not None:
        eng.set_error_model(err)   # +-2 band on every leaf; posterior evaluations fold it into the matrices
    full = eng.get_posterior(lam, mu, prior, per_family=True)
    yield dict(eng=eng, t=t, tree=tree, counts=counts, rng=rng, fr=fr, prior=prior, lam=lam, mu=mu, full=full, cfg=cfg,
               err=err, name=name)
    eng.close()


def test_score_is_sum_of_family_terms(problem):
    score, fz, ml, am, mp = problem["full"]
    assert fz == -1 and math.isfinite(score)
    assert score == pytest.approx(float(np.log(mp).sum()), rel=1e-12)
    assert np.all(ml > 0) and np.all((am >= 0) & (am < problem["rng"].root_max))


def test_family_values_do_not_depend_on_the_batch(problem):
    p = problem
    score, fz, ml, am, mp = p["full"]
    rs = np.random.RandomState(7)
    idx = np.sort(rs.choice(len(p["counts"]), 777, replace=False))
    p["eng"].set_families(p["counts"][idx], p["fr"])
    s2, fz2, ml2, am2, mp2 = p["eng"].get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    assert np.array_equal(ml2, ml[idx]) and np.array_equal(mp2, mp[idx]) and np.array_equal(am2, am[idx])
    p["eng"].set_families(p["counts"], p["fr"])


def test_compressed_walk_equals_the_uncompressed_one_at_full_size(problem):
    # subtree-state compression at the BASELINE shapes (not only on small trees): every per-family output of the whole
    # table, bit for bit, against the same library walking the full tree (option compress=0)
    import cafe_amd
    p = problem
    score, fz, ml, am, mp = p["full"]
    assert "used=1" in p["eng"].describe(), p["eng"].describe()
    eng = cafe_amd.Engine(0)
    try:
        eng.set_option("compress", 0)
        eng.set_tree(p["t"].parent, p["t"].left, p["t"].right, p["t"].branchlength)
        eng.set_families(p["counts"], p["fr"])
        if p["err"] is not None:
            eng.set_error_model(p["err"])
        s0, fz0, ml0, am0, mp0 = eng.get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
        assert "used=1" not in eng.describe()
    finally:
        eng.close()
    assert s0 == score and fz0 == fz
    assert np.array_equal(ml0, ml) and np.array_equal(mp0, mp) and np.array_equal(am0, am)


def _oracle_sample(counts, n, seed):
    """Rows compared with the oracle at full size (160 rows were 0.03 % of the 500 k table): n rows --
    the 64 with the largest single count, the 64 with the largest total, the 64 with the smallest total (the extremes
    are where an underflow, a clipped range or a wrong column limit would show first) and a uniform draw of the rest."""
    rs = np.random.RandomState(seed)
    big = np.argsort(counts.max(axis=1), kind="stable")[-64:]
    tot = np.argsort(counts.sum(axis=1), kind="stable")
    pick = set(big.tolist()) | set(tot[-64:].tolist()) | set(tot[:64].tolist())
    rest = rs.choice(len(counts), n, replace=False)
    for i in rest:
        if len(pick) >= n:
            break
        pick.add(int(i))
    return np.array(sorted(pick))


def test_oracle_spot_check_on_a_sample(problem):
    p = problem
    score, fz, ml, am, mp = p["full"]
    idx = _oracle_sample(p["counts"], 2048, 3)
    ekw = (dict(errormatrix=p["err"], err_mfs=p["rng"].max, leaf_has_err=np.ones(p["t"].n_nodes, np.uint8))
           if p["err"] is not None else {})
    so, fzo, mlo, amo, mpo = O.eval_posterior(p["t"], p["counts"][idx], p["rng"], p["lam"], p["mu"], p["prior"],
                                              nthreads=os.cpu_count() or 1, **ekw)
    assert np.max(np.abs(ml[idx] - mlo) / mlo) < 1e-9
    assert np.max(np.abs(mp[idx] - mpo) / mpo) < 1e-9
    assert np.array_equal(am[idx], amo)


def test_every_family_against_the_reference_arithmetic(problem):
    # Round 4: not a sample.  The library can run the reference's OWN arithmetic on the GPU -- exact-form matrices with the
    # host libm's exp() and a separate multiply and add per term (options k1=exact, k2=v1ref), pinned to the oracle BIT FOR
    # BIT (tests/test_gpu_reference_arithmetic.py, and on this table's oracle sample below).  Every family of the
    # full-size table is compared with that run: likelihood and posterior to 1e-9 relative (measured ~1e-12: the product
    # form of the search path), root argmax exactly.
    import cafe_amd
    import ctypes as C
    from cafe_amd import _lib
    p = problem
    score, fz, ml, am, mp = p["full"]
    eng = cafe_amd.Engine(0)
    try:
        eng.set_option("k1", "exact")
        eng.set_option("k2", "v1ref")
        eng.set_tree(p["t"].parent, p["t"].left, p["t"].right, p["t"].branchlength)
        eng.set_families(p["counts"], p["fr"])
        if p["err"] is not None:
            eng.set_error_model(p["err"])
        s1, fz1, ml1, am1, mp1 = eng.get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    finally:
        eng.close()
    assert fz1 == fz
    assert np.max(np.abs(ml - ml1) / ml1) < 1e-9 and np.max(np.abs(mp - mp1) / mp1) < 1e-9
    assert np.array_equal(am, am1)
    assert abs(score - s1) < 1e-9 * abs(s1)
    # ... and the reference-arithmetic run IS the oracle on the sample (bit for bit where the host's exp() is recognised)
    a, b = C.c_long(), C.c_long()
    recognised = _lib.load().cafehip_exp_like_host_selftest(200000, 5, C.byref(a), C.byref(b)) != 0
    idx = _oracle_sample(p["counts"], 512, 29)
    ekw = (dict(errormatrix=p["err"], err_mfs=p["rng"].max, leaf_has_err=np.ones(p["t"].n_nodes, np.uint8))
           if p["err"] is not None else {})
    so, fzo, mlo, amo, mpo = O.eval_posterior(p["t"], p["counts"][idx], p["rng"], p["lam"], p["mu"], p["prior"],
                                              nthreads=os.cpu_count() or 1, **ekw)
    if recognised:
        assert np.array_equal(ml1[idx], mlo)
    else:
        assert np.max(np.abs(ml1[idx] - mlo) / mlo) < 1e-12
    assert np.array_equal(am1[idx], amo)


def test_identity_error_model_is_a_no_op(problem):
    p = problem
    if p["err"] is not None:
        pytest.skip("this configuration carries its own error model")
    mfs = p["rng"].max
    p["eng"].set_error_model(np.eye(mfs + 1))
    try:
        sub = p["counts"][:2048]
        p["eng"].set_families(sub, p["fr"])
        s1, fz1, ml1, am1, mp1 = p["eng"].get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    finally:
        p["eng"].set_error_model(None)
    s0, fz0, ml0, am0, mp0 = p["eng"].get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    p["eng"].set_families(p["counts"], p["fr"])
    assert np.array_equal(ml1, ml0) and np.array_equal(mp1, mp0) and s1 == s0


def test_chunk_aligned_shards_combine_bit_exactly(problem):
    # the multi-GPU reduction on one GPU: each shard scored alone through the async API into device
    # buffers, chunk sums concatenated, fixed-order final sum == single-table score
    import torch
    from cafe_amd import distributed as D
    p = problem
    score = p["full"][0]
    F = len(p["counts"])
    for world in (2, 3, 8):
        bounds = D.shard_bounds(F, world)
        slots = D.max_chunks_per_rank(F, world)
        rows = []
        for lo, hi in bounds:
            p["eng"].set_families(p["counts"][lo:hi], p["fr"])
            packed, pc, pf = D.packed_buffer(torch, slots, "cuda")
            p["eng"].eval_posterior_async(p["lam"], p["mu"], p["prior"], pc, pf)
            torch.cuda.synchronize()
            rows.append(packed.cpu().numpy())
        host = np.stack(rows)
        fz_local = host[:, slots].copy().view(np.int32)[0::2]
        assert all(not (0 <= fz_local[r] < bounds[r][1] - bounds[r][0]) for r in range(world))
        assert D.final_score(host[:, :slots].reshape(-1), D.NO_ZERO) == score
    p["eng"].set_families(p["counts"], p["fr"])


# ---- BASELINE configs[4]: error model + Monte-Carlo null + report at full size ------------------------------------

def test_unfolded_error_model_equals_the_folded_one(problem):
    # the objective path folds the error model into the matrices once per evaluation (k1e_fold_error); with
    # option errfold=0 every family sums its band itself (cafe/cafe_tree.c:196-203 then :213-224): the two
    # must agree to rounding on a block of the table
    p = problem
    if p["err"] is None:
        pytest.skip("no error model in this configuration")
    sub = p["counts"][:4096]
    p["eng"].set_families(sub, p["fr"])
    folded = p["eng"].get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    p["eng"].set_option("errfold", 0)
    try:
        unfolded = p["eng"].get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    finally:
        p["eng"].set_option("errfold", 1)
        p["eng"].set_families(p["counts"], p["fr"])
    assert np.max(np.abs(folded[2] - unfolded[2]) / unfolded[2]) < 1e-12
    assert np.array_equal(folded[3], unfolded[3])
    assert np.array_equal(folded[2], p["full"][2][:4096])   # and a block equals the full-table values bit for bit


@pytest.fixture(scope="module")
def cfg5_report(tmp_path_factory):
    """configs[4] through the host driver at full size: 100 k families, error model on every leaf, lambda -s,
    Monte-Carlo null (250 root sizes x 1000 simulated families) and the report."""
    import torch
    torch.cuda.init()
    from cafe_amd import synth
    from cafe_amd.shell import CafeShell
    d = tmp_path_factory.mktemp("cfg5")
    tree, counts, cfg = synth.make_config("cfg5")
    rng = O.range_from_max(cfg["m"])
    tab, em, out = str(d / "families.tab"), str(d / "errormodel.txt"), str(d / "report")
    with open(tab, "w") as f:
        f.write("Desc\tFamily ID\t" + "\t".join(tree.leaf_names) + "\n")
        for i, row in enumerate(counts):
            f.write("NA\tF%06d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")
    synth.write_error_model_file(em, rng.max)
    sh = CafeShell(0, os.devnull)
    for line in ("seed 10", "tree " + cfg["newick"], "load -i %s -p 0.01 -t 1" % tab, "errormodel -model %s -all" % em,
                 "lambda -s"):
        sh.dispatch(line)
    lam_fit = float(sh.params[0])
    sh.dispatch("seed 10")                 # the null below starts from a known point of the rand() stream
    sh.dispatch("pvalue -o " + out + ".pv")
    sh.dispatch("report " + out)
    sh.close()
    null = np.array([[float(x) for x in l.split("\t")] for l in open(out + ".pv").read().splitlines()])
    return dict(tree=tree, counts=counts, cfg=cfg, rng=rng, lam_fit=lam_fit, null=null, report=out + ".cafe",
                t=O.PyTree(cfg["newick"]))


def test_cfg5_fitted_lambda_is_near_the_simulated_one(cfg5_report):
    r = cfg5_report
    assert r["null"].shape == (r["rng"].root_max - r["rng"].root_min + 1, 1000)
    # (not tight: the reference's Poisson prior fit starts Nelder-Mead at a uniform draw in [0, 1), where the one
    # forced count of 200 underflows poisspdf to 0 -- the fit stalls near its start, cafe/lambda.cpp:771-838 -- and
    # the resulting prior pulls the rate up; the restated driver reproduces that behaviour)
    assert abs(r["lam_fit"] - r["cfg"]["lam"]) < 0.5 * r["cfg"]["lam"]
    assert np.all(np.diff(r["null"], axis=1) >= 0)          # every root size's sample is sorted (:41)


def test_cfg5_monte_carlo_null_matches_the_oracle_on_root_sizes(cfg5_report):
    # the oracle replays the reference's `-t 1` rand() stream for ALL 250 x 1000 simulated families (cheap) and
    # scores those of a few root sizes (the dense CPU walk is the expensive part)
    r = cfg5_report
    t, rng = r["t"], r["rng"]
    lam = np.full(t.n_nodes, r["lam_fit"])
    mu = np.full(t.n_nodes, -1.0)
    mats = O.build_matrices(t, rng, lam, mu, nthreads=os.cpu_count() or 1)
    try:
        check = [rng.root_min, rng.root_min + 9, 57, rng.root_max]
        rows = O.mc_null_rows(t, rng, mats, 1000, 10, check)
        for s in check:
            cnt, cm = rows[s]
            lo = np.full(len(cm), s, np.int32)
            like = np.sort(O.eval_root_likelihoods(t, mats, cnt, lo, lo, cm, nthreads=os.cpu_count() or 1))
            got = r["null"][s - rng.root_min]
            ok = like > 0
            assert np.array_equal(got == 0, ~ok)
            assert np.max(np.abs(got[ok] - like[ok]) / like[ok]) < 2e-8      # the file holds 9 significant digits
    finally:
        O.free_matrices(mats)


def test_cfg5_report_sample_matches_the_oracle(cfg5_report):
    # Viterbi sizes, family-wide p-value and branch p-values of a sample of the 100 k report rows against the
    # oracle run on the same null distribution
    r = cfg5_report
    t, rng = r["t"], r["rng"]
    lam = np.full(t.n_nodes, r["lam_fit"])
    mu = np.full(t.n_nodes, -1.0)
    rep = O.parse_cafe_report(r["report"])
    assert len(rep) == len(r["counts"])
    mats = O.build_matrices(t, rng, lam, mu, nthreads=os.cpu_count() or 1)
    cd = np.ascontiguousarray(r["null"])
    # Newick order of the report's sizes -> node ids
    order = []

    def walk(v):
        if t.left[v] >= 0:
            walk(t.left[v])
            walk(t.right[v])
        order.append(v)

    walk(t.root)
    try:
        rs = np.random.RandomState(5)
        n_branch = 0
        for i in rs.choice(len(r["counts"]), 48, replace=False):
            sizes, maxp, pairs = rep["F%06d" % i]
            maxp_o, fs_o, bp_o = O.viterbi_and_pvalues(t, mats, r["counts"][i], cd, 1000, 0.01)
            got = np.zeros(t.n_nodes, np.int64)
            got[order] = sizes
            assert np.array_equal(got, fs_o), i
            assert abs(maxp - maxp_o) <= 2.5e-3, (i, maxp, maxp_o)          # %g text; a tie at 9 digits moves a rank by 1
            if bp_o is not None and all(x is not None for x in pairs):
                n_branch += 1
                flat = np.array([v for pr in pairs for v in pr])
                assert np.allclose(flat, bp_o, rtol=2e-5, atol=1e-300), i   # %g prints 6 significant digits
    finally:
        O.free_matrices(mats)


def test_cfg5_null_sharded_by_root_size_recombines_bit_exactly():
    # the multi-GPU split of the Monte-Carlo null (cafe/conditional_distribution.cpp:88-108: whole root sizes per
    # worker): blocks of root sizes scored separately equal the single 250,000-row launch bit for bit
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth
    tree, counts, cfg = synth.make_config("cfg5", F=512)
    rng = cafe_amd.init_family_size(cfg["m"])
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    lam, mu = synth.node_rates(tree, cfg)
    eng.set_exact_matrices(True)
    eng.reset_birthdeath_cache(lam, mu)
    mats = {v: eng.get_matrix(v) for v in range(tree.n_nodes) if v != tree.root}
    rows, lo, cm = synth.simulate_null_rows(tree, mats, rng, 1000, 123)
    R = rng.root_max - rng.root_min + 1
    whole = eng.eval_root_likelihoods(rows, lo, lo, cm)
    assert whole.shape == (R * 1000,) and np.all(np.isfinite(whole)) and np.count_nonzero(whole) > 0.9 * whole.size
    for world in (2, 3, 8):
        parts = []
        base, extra = divmod(R, world)
        r0 = 0
        for rank in range(world):
            r1 = r0 + base + (1 if rank < extra else 0)
            a, b = r0 * 1000, r1 * 1000
            parts.append(eng.eval_root_likelihoods(rows[a:b], lo[a:b], lo[a:b], cm[a:b]))
            r0 = r1
        assert np.array_equal(np.concatenate(parts), whole)
    # round 3: a tile's products stop at its largest column limit and its root step at its root sizes (option batch_trim);
    # the untrimmed launch of round 2 must give the same 250,000 values bit for bit, and trimming must save work
    trimmed_flops = eng.last_issued_flops()[0]
    eng.set_option("batch_trim", 0)
    untrimmed = eng.eval_root_likelihoods(rows, lo, lo, cm)
    untrimmed_flops = eng.last_issued_flops()[0]
    eng.set_option("batch_trim", 1)
    assert np.array_equal(untrimmed, whole)
    assert trimmed_flops < 0.75 * untrimmed_flops, (trimmed_flops, untrimmed_flops)
    eng.close()


# ---- BASELINE configs[3] at its stated size: 500 k families, 64 taxa, three lambda classes -------------------------

@pytest.fixture(scope="module")
def cfg4_500k(tmp_path_factory):
    """The 500k-family table as bench.py's strong-scaling leg builds it: 8 seeded blocks of 62,464 rows (generated once
    per test session and kept on disk)."""
    import torch
    torch.cuda.init()
    import cafe_amd
    from cafe_amd import synth
    from cafe_amd import tree as ctree
    cfg = dict(synth.CONFIGS["cfg4"])
    newick = synth.random_ultrametric_newick(cfg["n_taxa"], cfg["seed"])
    tree = ctree.CafeTree(newick)
    path = tmp_path_factory.getbasetemp() / "cfg4_500k.npy"
    if path.exists():
        counts = np.load(path)
    else:
        counts = np.concatenate([synth.simulate_families(tree, 62464, cfg["m"], cfg["lam"], cfg["mu"], cfg["seed"] + 1 + b)
                                 for b in range(8)])
        np.save(path, counts)
    rng = O.range_from_max(cfg["m"])
    lam, mu = synth.node_rates(tree, cfg)
    assert len(set(lam.tolist())) == 3
    return dict(tree=tree, t=O.PyTree(newick), counts=counts, rng=rng, lam=lam, mu=mu, cfg=cfg,
                fr=cafe_amd.FamilySizeRange(rng.min, rng.max, rng.root_min, rng.root_max), prior=O.prior_poisson(1000, rng.root_min, 8.0))


def test_configs3_500k_whole_table_8_way_split_and_oracle_sample(cfg4_500k):
    # the whole table on one context; the same table as 8 chunk-aligned blocks through the asynchronous entry point
    # (what 8 ranks compute), recombined with the fixed-order sum: bit-exact; per-family values of a block equal the
    # whole-table ones; 4,096 sampled families (extremes included) against the oracle under the three lambda classes
    import torch
    import cafe_amd
    from cafe_amd import distributed as D
    p = cfg4_500k
    F = len(p["counts"])
    assert F == 8 * 62464
    eng = cafe_amd.Engine(0)
    try:
        p["tree"].apply(eng)
        eng.set_families(p["counts"], p["fr"])
        score, fz, ml, am, mp = eng.get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
        assert fz == -1 and math.isfinite(score)
        assert score == pytest.approx(float(np.log(mp).sum()), rel=1e-12)
        bounds = D.shard_bounds(F, 8)
        slots = D.max_chunks_per_rank(F, 8)
        rows = []
        for r, (lo, hi) in enumerate(bounds):
            eng.set_families(p["counts"][lo:hi], p["fr"])
            packed, pc, pf = D.packed_buffer(torch, slots, "cuda")
            eng.eval_posterior_async(p["lam"], p["mu"], p["prior"], pc, pf)
            torch.cuda.synchronize()
            rows.append(packed.cpu().numpy())
            if r in (0, 5):
                s_b, fz_b, ml_b, am_b, mp_b = eng.get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
                assert np.array_equal(ml_b, ml[lo:hi]) and np.array_equal(mp_b, mp[lo:hi]) and np.array_equal(am_b, am[lo:hi])
        host = np.stack(rows)
        assert D.final_score(host[:, :slots].reshape(-1), D.NO_ZERO) == score
    finally:
        eng.close()
    idx = _oracle_sample(p["counts"], 4096, 17)
    so, fzo, mlo, amo, mpo = O.eval_posterior(p["t"], p["counts"][idx], p["rng"], p["lam"], p["mu"], p["prior"], nthreads=os.cpu_count() or 1)
    assert np.max(np.abs(ml[idx] - mlo) / mlo) < 1e-9
    assert np.max(np.abs(mp[idx] - mpo) / mpo) < 1e-9
    assert np.array_equal(am[idx], amo)
    # round 4: ALL 499,712 families against the reference's own arithmetic run on the GPU (k1=exact, k2=v1ref: pinned to the
    # oracle bit for bit, here on the same sample) -- the sample is no longer the only thing standing behind the other 99 %
    eng = cafe_amd.Engine(0)
    try:
        eng.set_option("k1", "exact")
        eng.set_option("k2", "v1ref")
        p["tree"].apply(eng)
        eng.set_families(p["counts"], p["fr"])
        s1, fz1, ml1, am1, mp1 = eng.get_posterior(p["lam"], p["mu"], p["prior"], per_family=True)
    finally:
        eng.close()
    assert fz1 == -1
    assert np.max(np.abs(ml - ml1) / ml1) < 1e-9 and np.max(np.abs(mp - mp1) / mp1) < 1e-9 and np.array_equal(am, am1)
    assert np.max(np.abs(ml1[idx] - mlo) / mlo) < 1e-12

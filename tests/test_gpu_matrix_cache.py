"""Matrices ahead of time (cafehip_prefetch_matrices, round 5): an evaluation that finds its parameter set in the device
store binds its nodes to matrices built earlier on the second stream and launches no matrix build.  The store is keyed like
the reference's cache -- (int branch length, lambda, mu) per node, doubles compared exactly (libtree/birthdeath.h:26-31,
cafe/cafe_tree.c:374-391) -- and the same kernel builds the matrices, so EVERY output of such an evaluation must equal,
bit for bit, the output of an evaluation that builds on demand."""
import os

import numpy as np
import pytest

from tests import _orc as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _table(config="cfg2", F=3000, error_model=False, options=()):
    import cafe_amd
    from cafe_amd import synth
    tree, counts, cfg = synth.make_config(config, F=F)
    rng = cafe_amd.init_family_size(cfg["m"])
    eng = cafe_amd.Engine(0)
    for k, v in options:
        eng.set_option(k, v)
    tree.apply(eng)
    eng.set_families(counts, rng)
    if error_model:
        eng.set_error_model(synth.banded_error_matrix(rng.max))
    prior = O.prior_poisson(1000, rng.root_min, 8.0)
    return eng, tree, rng, prior


def _sets(tree, n, with_mu=False, classes=False, seed=3):
    r = np.random.default_rng(seed)
    nl = np.empty((n, tree.n_nodes))
    nm = np.empty((n, tree.n_nodes))
    for i in range(n):
        if classes:
            cls = r.integers(0, 3, tree.n_nodes)
            lam = 0.001 + 0.002 * r.random(3)
            nl[i] = lam[cls]
        else:
            nl[i] = 0.001 + 0.002 * r.random()
        nm[i] = (0.5 + r.random()) * nl[i] if with_mu else -1.0
    return nl, nm


def _same(a, b):
    assert a[0] == b[0] and a[1] == b[1]
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("with_mu,classes,error_model", [(False, False, False), (True, False, False), (False, True, False), (False, False, True)])
def test_an_evaluation_on_prefetched_matrices_equals_one_that_builds_them(with_mu, classes, error_model):
    eng, tree, rng, prior = _table(error_model=error_model)
    ref, _, _, _ = _table(error_model=error_model, options=(("matrix_cache", 0),))
    nl, nm = _sets(tree, 5, with_mu, classes)
    want = [ref.get_posterior(nl[i], nm[i], prior, per_family=True) for i in range(5)]
    # the first evaluation builds on demand (it also puts the prior on the device); the others are announced first
    _same(eng.get_posterior(nl[0], nm[0], prior, per_family=True), want[0])
    eng.prefetch_matrices(nl[1:4], nm[1:4])
    for i in (2, 1, 3):
        _same(eng.get_posterior(nl[i], nm[i], prior, per_family=True), want[i])
        # the matrices the evaluation was bound to are the ones a demand build produces
        node = 2
        assert np.array_equal(eng.get_matrix(node), _matrix_after(ref, nl[i], nm[i], prior, node))
    st = eng.matrix_cache_stats()
    assert st["hits"] == 3 and st["built"] == 3, st
    # a set that was never announced is built on demand, and the store is untouched by it
    _same(eng.get_posterior(nl[4], nm[4], prior, per_family=True), want[4])
    _same(eng.get_posterior(nl[2], nm[2], prior, per_family=True), want[2])
    st = eng.matrix_cache_stats()
    assert st["hits"] == 4 and st["misses"] >= 1, st
    ref_st = ref.matrix_cache_stats()
    assert ref_st["entries"] == 0 and ref_st["hits"] == 0
    eng.close()
    ref.close()


def _matrix_after(eng, nl, nm, prior, node):
    eng.get_posterior(nl, nm, prior)
    return eng.get_matrix(node)


def test_request_parked_behind_the_next_evaluation_and_replacement_of_the_least_recently_used():
    eng, tree, rng, prior = _table(options=(("matrix_cache", 4),))
    ref, _, _, _ = _table(options=(("matrix_cache", 0),))
    nl, nm = _sets(tree, 12, seed=11)
    want = [ref.get_posterior(nl[i], nm[i], prior) for i in range(12)]
    assert eng.get_posterior(nl[0], nm[0], prior) == want[0]
    # as a search does: before evaluation i, announce what may follow it
    for i in range(1, 11):
        eng.prefetch_matrices(nl[i + 1:i + 2], nm[i + 1:i + 2], when=eng.PREFETCH_BEHIND_NEXT_EVALUATION)
        assert eng.get_posterior(nl[i], nm[i], prior) == want[i], i
    st = eng.matrix_cache_stats()
    assert st["entries"] == 4
    assert st["hits"] == 9, st          # evaluations 2..10 were announced one evaluation ahead
    assert st["replaced"] >= 5, st      # ten sets went through four entries
    # more sets than entries in one request: the surplus is dropped, nothing breaks, the bound entry survives
    eng.prefetch_matrices(nl[:8], nm[:8])
    for i in (10, 0, 1, 2, 7):
        assert eng.get_posterior(nl[i], nm[i], prior) == want[i], i
    eng.close()
    ref.close()


def test_what_invalidates_the_store():
    eng, tree, rng, prior = _table()
    nl, nm = _sets(tree, 3, seed=5)
    base = eng.get_posterior(nl[0], nm[0], prior)
    eng.prefetch_matrices(nl, nm)
    # another prior: the evaluation must go through K1 (which mirrors the prior), not through the store
    prior2 = O.prior_poisson(1000, rng.root_min, 5.0)
    hits0 = eng.matrix_cache_stats()["hits"]
    s2 = eng.get_posterior(nl[1], nm[1], prior2)
    assert eng.matrix_cache_stats()["hits"] == hits0
    assert s2 != eng.get_posterior(nl[1], nm[1], prior)
    # exact-form matrices are other matrices: entries built in the product form must not serve them
    eng.prefetch_matrices(nl, nm)
    eng.set_exact_matrices(True)
    hits0 = eng.matrix_cache_stats()["hits"]
    exact = eng.get_posterior(nl[2], nm[2], prior, per_family=True)
    assert eng.matrix_cache_stats()["hits"] == hits0
    eng.prefetch_matrices(nl[2:3], nm[2:3])     # ... built in the exact form now
    _same(eng.get_posterior(nl[2], nm[2], prior, per_family=True), exact)
    assert eng.matrix_cache_stats()["hits"] == hits0 + 1
    eng.set_exact_matrices(False)
    assert eng.get_posterior(nl[0], nm[0], prior) == base
    # a new table of another matrix side: entries are laid out again
    import cafe_amd
    from cafe_amd import synth
    tree3, counts3, cfg3 = synth.make_config("cfg3", F=600)
    rng3 = cafe_amd.init_family_size(cfg3["m"])
    tree3.apply(eng)
    eng.set_families(counts3, rng3)
    prior3 = O.prior_poisson(1000, rng3.root_min, 8.0)
    nl3, nm3 = _sets(tree3, 2, with_mu=True, seed=9)
    a = eng.get_posterior(nl3[0], nm3[0], prior3, per_family=True)
    eng.prefetch_matrices(nl3, nm3)
    _same(eng.get_posterior(nl3[0], nm3[0], prior3, per_family=True), a)
    eng.close()


def test_rates_the_product_form_cannot_take_are_left_to_the_demand_build():
    # lambda * t >= 1 gives a zero matrix (mode 0), tiny rates leave the product form's range: whatever the mix, values agree
    eng, tree, rng, prior = _table(F=800)
    ref, _, _, _ = _table(F=800, options=(("matrix_cache", 0),))
    nl = np.array([np.full(tree.n_nodes, x) for x in (0.2, 1e-9, 0.0021, 3e-5)])
    nm = np.full_like(nl, -1.0)
    want = [ref.get_posterior(nl[i], nm[i], prior, per_family=True) for i in range(4)]
    eng.get_posterior(nl[2], nm[2], prior)
    eng.prefetch_matrices(nl, nm)
    for i in range(4):
        got = eng.get_posterior(nl[i], nm[i], prior, per_family=True)
        assert (got[0] == want[i][0] or (np.isinf(got[0]) and np.isinf(want[i][0]))) and got[1] == want[i][1]
        for x, y in zip(got[2:], want[i][2:]):
            assert np.array_equal(x, y)
    eng.close()
    ref.close()


def test_report_phase_calls_use_the_bound_matrices():
    # the batch entry points read "the matrices of the last evaluation": after an evaluation served from the store those are
    # the entry's
    eng, tree, rng, prior = _table(F=500)
    ref, _, _, _ = _table(F=500, options=(("matrix_cache", 0),))
    nl, nm = _sets(tree, 2, seed=21)
    for e in (eng, ref):
        e.get_posterior(nl[0], nm[0], prior)
    eng.prefetch_matrices(nl[1:], nm[1:])
    eng.get_posterior(nl[1], nm[1], prior)
    ref.get_posterior(nl[1], nm[1], prior)
    assert eng.matrix_cache_stats()["hits"] == 1
    rows = np.random.default_rng(1).integers(0, 30, (64, (tree.n_nodes + 1) // 2)).astype(np.int32)
    lo = np.full(64, rng.root_min, np.int32)
    hi = np.full(64, rng.root_max, np.int32)
    cm = np.full(64, rng.max, np.int32)
    assert np.array_equal(eng.eval_root_likelihoods(rows, lo, hi, cm), ref.eval_root_likelihoods(rows, lo, hi, cm))
    eng.close()
    ref.close()


# ---- the searches of the host driver: look-ahead on / off ---------------------------------------------------------------
NEWICK = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"


def _search(lines, lookahead):
    from cafe_amd.shell import CafeShell
    sh = CafeShell(0, os.devnull)
    sh.set_option("speculate", 0)        # (whole-evaluation batching would serve the small example table instead)
    sh.set_option("lookahead", 1 if lookahead else 0)
    for l in lines:
        sh.dispatch(l)
    res = (list(sh.params), sh.score, sh.iterations, sh.evaluations, sh.trace().tolist())
    stats = sh.lookahead_stats()
    secs = sh.search_seconds
    sh.close()
    return res, stats, secs


@pytest.mark.parametrize("command", ["lambda -s", "lambdamu -s", "lambda -s -t (((1,1)1,(2,2)2)2,2)"])
def test_matrices_ahead_of_time_leave_the_search_trajectory_unchanged(command):
    lines = ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + NEWICK, command]
    plain, st0, t0 = _search(lines, False)
    ahead, st1, t1 = _search(lines, True)
    assert ahead == plain                        # parameters, score, iterations, evaluations, every objective call
    assert st0["announcements"] == 0 and st0["hits"] == 0
    n_eval = plain[3]
    # one parameter: every evaluation but the first of a search can be foreseen; more parameters: the evaluations that
    # follow a reflection, an expansion or a contraction (not the ones after the initial simplex or a shrink)
    need = 0.9 if command == "lambda -s" else 0.6
    assert st1["hits"] >= need * (n_eval - 1), (st1, n_eval)
    print("%s: %d evaluations, %d served from matrices built ahead (%d sets built in %d announcements); search %.2f -> %.2f ms"
          % (command, n_eval, st1["hits"], st1["built"], st1["announcements"], 1e3 * t0, 1e3 * t1))


def test_a_search_on_a_table_that_fills_the_chip_with_an_error_model():
    # 3,000 families under the 16-taxon tree, error model on: look-ahead serves folded matrices too
    import cafe_amd
    from cafe_amd import synth
    from cafe_amd.shell import CafeShell
    import tempfile
    tree, counts, cfg = synth.make_config("cfg2", F=3000)
    with tempfile.TemporaryDirectory() as d:
        tab = os.path.join(d, "t.tab")
        em = os.path.join(d, "errormodel.txt")
        with open(tab, "w") as f:
            f.write("Desc\tFamily ID\t" + "\t".join(tree.leaf_names) + "\n")
            for i, row in enumerate(counts):
                f.write("NA\tF%06d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")
        synth.write_error_model_file(em, cafe_amd.init_family_size(cfg["m"]).max)
        lines = ["seed 4", "tree " + cfg["newick"], "load -i %s" % tab, "errormodel -model %s -all" % em, "lambda -s"]
        out = []
        for look in (0, 1):
            sh = CafeShell(0, os.devnull)
            sh.set_option("lookahead", look)
            for l in lines:
                sh.dispatch(l)
            out.append(((list(sh.params), sh.score, sh.iterations, sh.evaluations, sh.trace().tolist()), sh.lookahead_stats()))
            sh.close()
    assert out[0][0] == out[1][0]
    assert out[1][1]["hits"] >= 0.9 * (out[1][0][3] - 1), out[1][1]


def test_moving_the_matrix_storage_forgets_the_matrices_built_on_demand():
    """resizing the store of sets built ahead (option matrix_cache) reallocates and zero-fills the matrix buffer; the
    matrices an ordinary evaluation left there are gone, so cafehip_get_matrix must refuse (or rebuild), never hand out zeros."""
    eng, tree, rng, prior = _table(options=(("matrix_cache", 0),))
    nl, nm = _sets(tree, 1)
    eng.get_posterior(nl[0], nm[0], prior)
    before = eng.get_matrix(2)
    assert before.sum() > 0
    eng.set_option("matrix_cache", 4)
    try:
        after = eng.get_matrix(2)
    except RuntimeError:
        after = None
    assert after is None or np.array_equal(after, before)
    # and the next evaluation rebuilds: same matrix again
    eng.get_posterior(nl[0], nm[0], prior)
    assert np.array_equal(eng.get_matrix(2), before)
    eng.close()


def test_a_set_left_to_the_demand_build_does_not_evict_an_entry():
    """a candidate whose keys the launch's arithmetic form cannot take (a rate so small that rho^8 leaves the double
    range: per-term keys in a product-form launch) is skipped by the store -- and must not cost a valid entry its slot: with the
    store FULL of valid sets, announcing only such a set evicts nothing and every stored set still hits."""
    eng, tree, rng, prior = _table(F=800, options=(("matrix_cache", 4),))
    good = [np.full(tree.n_nodes, x) for x in (0.0019, 0.0021, 0.0023, 0.0025)]
    mu = np.full(tree.n_nodes, -1.0)
    eng.get_posterior(good[0], mu, prior)
    eng.prefetch_matrices(np.array(good), np.array([mu] * 4))
    for g in good:
        eng.get_posterior(g, mu, prior)
    before = eng.matrix_cache_stats()
    # lambda / mu with a denormal mu: alpha ~ 1e-320, coeff / (alpha * beta) is not finite -> no product form for this set
    eng.prefetch_matrices(np.array([np.full(tree.n_nodes, 0.002)]), np.array([np.full(tree.n_nodes, 1e-320)]))
    after = eng.matrix_cache_stats()
    assert after["replaced"] == before["replaced"], (before, after)
    for g in good:
        eng.get_posterior(g, mu, prior)
    last = eng.matrix_cache_stats()
    assert last["hits"] - after["hits"] >= 3, (after, last)   # (the bound entry's own set may be rebuilt on demand)
    eng.close()

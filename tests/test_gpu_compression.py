"""Subtree-state compression of the objective path (cafe_amd/csrc/schedule.hpp, CTile): families that agree on the
counts below a node share its vector, so the product with the node's edge matrix is built once per distinct state
and the family walk gathers it.  The values must be BIT-identical to the uncompressed walk (same products, same
order), with and without a folded error model, for one and several parameter sets."""
import os

import numpy as np
import pytest

from tests import _orc as O

pytestmark = pytest.mark.gpu

NEWICK = "(((a:6,b:6):5,(c:4,(d:2,e:2):2):7):9,((f:3,g:3):8,h:11):9)"


def _table(F, n, seed, top=9):
    rs = np.random.RandomState(seed)
    counts = rs.poisson(2.2, size=(F, n)).clip(0, top).astype(np.int32)
    counts[0] = 0
    counts[1] = top
    return counts


def _run(compress, counts, t, rng_tuple, lam, mu, prior, err=None, sets=None, err_leaves=None):
    import cafe_amd
    if True:
        eng = cafe_amd.Engine(0)
        try:
            eng.set_option("compress", 1 if compress else 0)
            eng.set_tree(t.parent, t.left, t.right, t.branchlength)
            eng.set_families(counts, cafe_amd.FamilySizeRange(*rng_tuple))
            if err is not None:
                eng.set_error_model(err, err_leaves)
            if sets is None:
                out = eng.get_posterior(lam, mu, prior, per_family=True)
            else:
                out = eng.get_posterior_multi(sets[0], sets[1], prior)
            return out, eng.describe(), eng.last_issued_flops()
        finally:
            eng.close()


@pytest.mark.parametrize("model", ["lambda", "lambdamu"])
@pytest.mark.parametrize("with_error", [False, True])
def test_compressed_walk_is_bit_identical_and_matches_the_oracle(model, with_error):
    t = O.PyTree(NEWICK)
    F = 6000
    counts = _table(F, t.n_leaves, 11)
    rng_tuple = (0, 40, 1, 30)
    rng = O.make_range(*rng_tuple)
    prior = O.prior_poisson(1000, 1, 2.0)
    lam = np.full(t.n_nodes, 0.02)
    mu = np.full(t.n_nodes, 0.013 if model == "lambdamu" else -1.0)
    err = None
    if with_error:
        from cafe_amd import synth
        err = synth.banded_error_matrix(rng_tuple[1])
    (s1, fz1, ml1, am1, mp1), d1, (w1, tb1) = _run(True, counts, t, rng_tuple, lam, mu, prior, err)
    (s0, fz0, ml0, am0, mp0), d0, (w0, tb0) = _run(False, counts, t, rng_tuple, lam, mu, prior, err)
    assert "compressed(" in d1 and "used=1" in d1, d1
    assert "compressed(" not in d0
    assert tb1 > 0 and tb0 == 0 and w1 + tb1 < w0, (w1, tb1, w0)      # less matrix work, same values:
    assert s1 == s0 and fz1 == fz0
    assert np.array_equal(ml1, ml0) and np.array_equal(mp1, mp0) and np.array_equal(am1, am0)
    if not with_error:
        so, fzo, mlo, amo, mpo = O.eval_posterior(t, counts, rng, lam, mu, prior, nthreads=os.cpu_count() or 1)
        nz = mlo > 0
        assert np.max(np.abs(ml1[nz] - mlo[nz]) / mlo[nz]) < 1e-9
        assert np.max(np.abs(mp1[nz] - mpo[nz]) / mpo[nz]) < 1e-9
        assert abs(s1 - so) <= 1e-9 * abs(so)


def test_compressed_walk_with_several_parameter_sets():
    t = O.PyTree(NEWICK)
    counts = _table(5000, t.n_leaves, 12)
    rng_tuple = (0, 40, 1, 30)
    prior = O.prior_poisson(1000, 1, 2.0)
    K = 3
    lams = np.stack([np.full(t.n_nodes, 0.01 * (k + 1)) for k in range(K)])
    mus = np.full((K, t.n_nodes), -1.0)
    out1, d1, _ = _run(True, counts, t, rng_tuple, None, None, prior, sets=(lams, mus))
    out0, d0, _ = _run(False, counts, t, rng_tuple, None, None, prior, sets=(lams, mus))
    assert "used=1" in d1, d1
    for x, y in zip(out1, out0):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def test_small_tables_and_incompressible_tables_are_left_alone():
    import cafe_amd
    t = O.PyTree(NEWICK)
    eng = cafe_amd.Engine(0)
    try:
        eng.set_tree(t.parent, t.left, t.right, t.branchlength)
        eng.set_families(_table(40, t.n_leaves, 13), cafe_amd.FamilySizeRange(0, 40, 1, 30))
        assert "compressed(" not in eng.describe()
        rs = np.random.RandomState(5)
        wide = rs.randint(0, 200, size=(3000, t.n_leaves)).astype(np.int32)     # every pair of counts distinct
        eng.set_families(wide, cafe_amd.FamilySizeRange(0, 260, 1, 250))
        assert "compressed(" not in eng.describe()
    finally:
        eng.close()


def test_error_model_on_some_species_only_and_root_with_two_compressed_children():
    """The folded matrix is used for exactly the leaves that carry the model, inside compressed subtrees (k2c_nodes
    reads the per-leaf flag by count-table column) and above them (the walk's flag is per walk column); here every
    non-root node compresses, so the walk is the root step alone, gathering two table rows."""
    from cafe_amd import synth
    t = O.PyTree("((a:6,b:6):5,(c:4,d:7):3)")
    counts = _table(4000, t.n_leaves, 21, top=30)
    rng_tuple = (0, 60, 1, 40)
    rng = O.make_range(*rng_tuple)
    prior = O.prior_poisson(1000, 1, 2.0)
    lam = np.full(t.n_nodes, 0.02)
    mu = np.full(t.n_nodes, -1.0)
    err = synth.banded_error_matrix(rng_tuple[1])
    leaves = np.zeros(t.n_nodes, np.uint8)
    leaves[0] = leaves[4] = 1          # nodes 0, 2, 4, 6 are the leaves a, b, c, d: the model on a and c only
    (s1, fz1, ml1, am1, mp1), d1, _ = _run(True, counts, t, rng_tuple, lam, mu, prior, err, err_leaves=leaves)
    (s0, fz0, ml0, am0, mp0), d0, _ = _run(False, counts, t, rng_tuple, lam, mu, prior, err, err_leaves=leaves)
    assert "walk_steps=1" in d1 and "used=1" in d1, d1
    assert s1 == s0 and np.array_equal(ml1, ml0) and np.array_equal(mp1, mp0) and np.array_equal(am1, am0)
    E = np.asarray(err)
    so, fzo, mlo, amo, mpo = O.eval_posterior(t, counts, rng, lam, mu, prior, nthreads=os.cpu_count() or 1,
                                              errormatrix=E, err_mfs=E.shape[0] - 1, leaf_has_err=leaves)
    nz = mlo > 0
    assert np.max(np.abs(ml1[nz] - mlo[nz]) / mlo[nz]) < 1e-9
    assert abs(s1 - so) <= 1e-9 * abs(so)


def test_plan_follows_tables_trees_and_error_models_through_one_context():
    """One context through a sequence of tables, trees and error models (what lhtest / a scripted session does): the
    compression plan is rebuilt by set_families / set_tree and its error-model flags by set_error_model; every
    evaluation equals the one of a context that never compresses."""
    import cafe_amd
    from cafe_amd import synth
    trees = [O.PyTree(NEWICK), O.PyTree("((a:3,(b:5,c:5):2):4,((d:1,e:9):6,(f:2,(g:4,h:4):7):3):2)")]
    rng_tuple = (0, 40, 1, 30)
    prior = O.prior_poisson(1000, 1, 2.0)
    err = synth.banded_error_matrix(rng_tuple[1])
    steps = [  # (tree, rows, seed, error model, lambda)
        (0, 5000, 1, None, 0.02), (0, 50, 2, None, 0.02), (1, 50, 2, None, 0.015), (1, 7000, 3, None, 0.015),
        (1, 7000, 3, err, 0.015), (0, 7000, 3, err, 0.03), (0, 7000, 3, None, 0.03), (0, 2500, 4, None, 0.01),
    ]

    def session(compress):
        out = []
        if True:
            eng = cafe_amd.Engine(0)
            eng.set_option("compress", 1 if compress else 0)
            try:
                cur = (None, None, None, None)
                for ti, rows, seed, e, lam in steps:
                    t = trees[ti]
                    if cur[0] != ti:
                        eng.set_tree(t.parent, t.left, t.right, t.branchlength)
                    if cur[1:3] != (rows, seed):
                        eng.set_families(_table(rows, t.n_leaves, seed), cafe_amd.FamilySizeRange(*rng_tuple))
                    if (cur[3] is None) != (e is None) or cur[0] != ti:
                        eng.set_error_model(e)
                    cur = (ti, rows, seed, e)
                    r = eng.get_posterior(np.full(t.n_nodes, lam), np.full(t.n_nodes, -1.0), prior, per_family=True)
                    out.append((r, "used=1" in eng.describe()))
            finally:
                eng.close()
        return out

    a, b = session(True), session(False)
    used = [u for _, u in a]
    assert used == [True, False, False, True, True, True, True, True], used
    for (ra, _), (rb, _) in zip(a, b):
        assert ra[0] == rb[0] and ra[1] == rb[1]
        for x, y in zip(ra[2:], rb[2:]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("max_size", [40, 70, 100])   # matrix sides 41 / 71 / 101 -> 3 / 5 / 7 row tiles (odd: a lone last tile)
@pytest.mark.parametrize("with_error", [False, True])
def test_paired_row_tiles_of_the_table_kernel_change_no_bit(max_size, with_error):
    """Round 5: on levels of many tiles k2c_nodes deals a wave two row tiles and reads both with one 16-byte load per
    k-step (option k2c_pair: -1 by level size, 0 never, 1 always).  Same operands in the same order per accumulator:
    forced on and forced off must agree bit for bit, for even and odd numbers of row tiles, with and without an error model,
    for one parameter set and for several."""
    import cafe_amd
    t = O.PyTree(NEWICK)
    counts = _table(5000, t.n_leaves, 5, top=min(max_size - 1, 25))
    rng_tuple = (0, max_size, 1, max_size - 10)
    prior = O.prior_poisson(1000, 1, 2.0)
    lam = np.full(t.n_nodes, 0.015)
    mu = np.full(t.n_nodes, 0.011)
    sets = (np.stack([lam, 1.3 * lam, 0.6 * lam]), np.stack([mu, mu, 1.2 * mu]))
    outs = {}
    for pair in (0, 1):
        eng = cafe_amd.Engine(0)
        try:
            eng.set_option("k2c_pair", pair)
            eng.set_tree(t.parent, t.left, t.right, t.branchlength)
            eng.set_families(counts, cafe_amd.FamilySizeRange(*rng_tuple))
            if with_error:
                from cafe_amd import synth
                eng.set_error_model(synth.banded_error_matrix(max_size), None)
            one = eng.get_posterior(lam, mu, prior, per_family=True)
            many = eng.get_posterior_multi(sets[0], sets[1], prior)
            assert "compressed(" in eng.describe()
            outs[pair] = (one, many)
        finally:
            eng.close()
    (s0, fz0, ml0, am0, mp0), m0 = outs[0]
    (s1, fz1, ml1, am1, mp1), m1 = outs[1]
    assert s0 == s1 and fz0 == fz1
    assert np.array_equal(ml0, ml1) and np.array_equal(am0, am1) and np.array_equal(mp0, mp1)
    for x, y in zip(m0, m1):
        assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("options", [
    {"k2c_gemm": 0},                                   # round 5's k2c_nodes
    {"k2c_gemm": 1, "k2c_nst": 1, "k2c_pair": 0},      # k2c_gemm (round 6): every state-tile count, one and two row tiles per wave,
    {"k2c_gemm": 1, "k2c_nst": 1, "k2c_pair": 1},      # dispatch and XCD-aware tile order
    {"k2c_gemm": 1, "k2c_nst": 2, "k2c_pair": 0},
    {"k2c_gemm": 1, "k2c_nst": 2, "k2c_pair": 1, "k2c_xcd": 0},
    {"k2c_gemm": 1, "k2c_nst": 4, "k2c_pair": 0},
    {"k2c_gemm": 1, "k2c_nst": 4, "k2c_pair": 1},
])
@pytest.mark.parametrize("shape", ["narrow", "wide", "error"])
def test_every_table_kernel_shape_builds_the_same_tables(options, shape):
    """The factor tables as a chunk-pipelined GEMM (k2c_gemm.hpp) -- 16, 32 or 64 states per workgroup, matrices of 3 / 5 / 7 row
    tiles with a partial last chunk of every length -- against the uncompressed walk: every per-family output ==."""
    import cafe_amd
    from cafe_amd import synth
    t = O.PyTree(NEWICK)
    top, rmax = {"narrow": (9, 40), "wide": (30, 104), "error": (12, 70)}[shape]
    counts = _table(7000, t.n_leaves, 21, top=top)
    rng_tuple = (0, rmax, 1, 30)
    prior = O.prior_poisson(1000, 1, 2.0)
    lam = np.full(t.n_nodes, 0.02)
    mu = np.full(t.n_nodes, 0.013 if shape == "wide" else -1.0)
    err = synth.banded_error_matrix(rng_tuple[1]) if shape == "error" else None
    (s0, fz0, ml0, am0, mp0), d0, _ = _run(False, counts, t, rng_tuple, lam, mu, prior, err)
    eng = cafe_amd.Engine(0)
    try:
        for k, v in options.items():
            eng.set_option(k, v)
        eng.set_tree(t.parent, t.left, t.right, t.branchlength)
        eng.set_families(counts, cafe_amd.FamilySizeRange(*rng_tuple))
        if err is not None:
            eng.set_error_model(err)
        s1, fz1, ml1, am1, mp1 = eng.get_posterior(lam, mu, prior, per_family=True)
        d1 = eng.describe()
        # the option can also be changed with the table in place: the plan is rebuilt
        eng.set_option("k2c_nst", 0)
        s2, fz2, ml2, am2, mp2 = eng.get_posterior(lam, mu, prior, per_family=True)
    finally:
        eng.close()
    assert "compressed(" in d1 and "used=1" in d1, d1
    assert s1 == s0 and fz1 == fz0 and s2 == s0
    assert np.array_equal(ml1, ml0) and np.array_equal(mp1, mp0) and np.array_equal(am1, am0)
    assert np.array_equal(ml2, ml0) and np.array_equal(mp2, mp0)

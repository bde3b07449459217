"""Batch mode (cafehip_eval_root_likelihoods: per-row root range and column limit -- the Monte-Carlo null,
cafe/conditional_distribution.cpp:16-32, and the per-family ranges of cafe/cafe_family.c:236-255) with the round-3
trimming: a workgroup's products stop at the largest column limit of its rows and its root step covers only the row
tiles that hold a root size one of its rows asks for.  Rows beyond a row's limit are zero in the reference's
arithmetic, so the trimmed launch must equal the untrimmed one BIT FOR BIT -- for limits mixed inside a tile, root
ranges of one size and of many, a ragged last tile, both matrix-instruction shapes -- and the oracle to 1e-9."""
import os

import numpy as np
import pytest

from tests import _orc as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["4", "16"])
@pytest.mark.parametrize("newick,mx", [("((a:7,b:11):5,(c:3,(d:9,e:2):6):4)", 37),
                                       ("(((a:3,b:3):4,(c:5,d:5):2):6,((e:1,f:1):8,(g:2,(h:1,i:1):1):7):4)", 90)])
def test_trimmed_batch_equals_untrimmed_and_oracle(shape, newick, mx):
    import cafe_amd
    t = O.PyTree(newick)
    rs = np.random.RandomState(11 + mx)
    rmax = max(2, int(mx * 0.8))
    rng = O.make_range(0, mx, 1, rmax)
    F = 64
    table = rs.randint(0, 6, size=(F, t.n_leaves)).astype(np.int32)
    table[0, 0] = mx
    B = 1000 + 37                                   # ragged last tile
    rows = rs.randint(0, mx // 2, size=(B, t.n_leaves)).astype(np.int32)
    col_max = rs.randint(3, mx + 1, size=B).astype(np.int32)
    col_max[:300] = np.sort(col_max[:300])          # a run of similar limits (what a Monte-Carlo null looks like) ...
    col_max[300:320] = 3                            # ... tiles of tiny limits, rows whose counts exceed them ...
    col_max[500] = mx                               # ... and one full-width row inside an otherwise narrow tile
    lo = rs.randint(1, rmax + 1, size=B).astype(np.int32)
    hi = lo.copy()
    wide = rs.rand(B) < 0.2                          # per-family ranges: several root sizes
    hi[wide] = np.minimum(lo[wide] + rs.randint(0, 9, size=wide.sum()), rmax)
    lam = np.full(t.n_nodes, 0.02)
    mu = np.full(t.n_nodes, -1.0)
    eng = cafe_amd.Engine(0)
    try:
        eng.set_option("mfma", shape)
        eng.set_tree(t.parent, t.left, t.right, t.branchlength)
        eng.set_families(table, cafe_amd.FamilySizeRange(0, mx, 1, rmax))
        eng.reset_birthdeath_cache(lam, mu)
        trimmed = eng.eval_root_likelihoods(rows, lo, hi, col_max)
        f1 = eng.last_issued_flops()[0]
        eng.set_option("batch_trim", 0)
        plain = eng.eval_root_likelihoods(rows, lo, hi, col_max)
        f0 = eng.last_issued_flops()[0]
        desc = eng.describe()
    finally:
        eng.close()
    assert ("mfma4x4" in desc) == (shape == "4"), desc
    assert np.array_equal(trimmed, plain), desc
    assert f1 < f0
    mats = O.build_matrices(t, rng, lam, mu, nthreads=2)
    try:
        ref = O.eval_root_likelihoods(t, mats, rows, lo, hi, col_max, nthreads=os.cpu_count() or 1)
    finally:
        O.free_matrices(mats)
    nz = ref > 0
    assert np.array_equal(trimmed == 0, ~nz)
    assert np.max(np.abs(trimmed[nz] - ref[nz]) / ref[nz]) < 1e-9


@pytest.mark.parametrize("shape,grid", [("16", "1,5,1,2"), ("16", "1,7,1,2"), ("16", "2,3,1,4"), ("16", "1,3,2,4"),
                                        ("4", "2,5,1,2"), ("4", "1,7,1,2"), ("4", "3,4,1,3"), ("4", "5,3,2,4")])
def test_every_live_row_tile_count_of_a_wide_wave_tile(shape, grid):
    """A trimmed tile deals 1 .. NRT_W row tiles to a wave and every count has its own instantiation of the product
    (mfma_edge_few): wave tiles up to 7 row tiles wide on a 151-wide matrix, column limits from 3 to 150 in sorted runs
    so that every count occurs -- bit for bit against the untrimmed launch of the same grid, and against the oracle."""
    import cafe_amd
    t = O.PyTree("(((a:3,b:3):4,(c:5,d:5):2):6,((e:1,f:1):8,(g:2,(h:1,i:1):1):7):4)")
    mx, rmax = 150, 120
    rng = O.make_range(0, mx, 1, rmax)
    rs = np.random.RandomState(5)
    table = rs.randint(0, 6, size=(64, t.n_leaves)).astype(np.int32)
    table[0, 0] = mx
    B = 640 + 9
    col_max = np.sort(rs.randint(3, mx + 1, size=B)).astype(np.int32)
    col_max[-40:] = rs.randint(3, mx + 1, size=40)           # and a few tiles of mixed limits
    rows = np.minimum(rs.randint(0, 12, size=(B, t.n_leaves)), col_max[:, None]).astype(np.int32)
    lo = rs.randint(1, rmax + 1, size=B).astype(np.int32)
    hi = np.minimum(lo + (rs.rand(B) < 0.2) * rs.randint(0, 20, size=B), rmax).astype(np.int32)
    lam = np.full(t.n_nodes, 0.02)
    mu = np.full(t.n_nodes, 0.013)
    eng = cafe_amd.Engine(0)
    try:
        eng.set_option("mfma", shape)
        eng.set_option("k2cfg" if shape == "16" else "k2cfg4", grid)
        eng.set_tree(t.parent, t.left, t.right, t.branchlength)
        eng.set_families(table, cafe_amd.FamilySizeRange(0, mx, 1, rmax))
        eng.reset_birthdeath_cache(lam, mu)
        trimmed = eng.eval_root_likelihoods(rows, lo, hi, col_max)
        desc = eng.describe()
        eng.set_option("batch_trim", 0)
        plain = eng.eval_root_likelihoods(rows, lo, hi, col_max)
    finally:
        eng.close()
    assert "=%s " % grid in desc, desc
    assert np.array_equal(trimmed, plain), desc
    mats = O.build_matrices(t, rng, lam, mu, nthreads=2)
    try:
        ref = O.eval_root_likelihoods(t, mats, rows, lo, hi, col_max, nthreads=os.cpu_count() or 1)
    finally:
        O.free_matrices(mats)
    nz = ref > 0
    assert np.array_equal(trimmed == 0, ~nz)
    assert np.max(np.abs(trimmed[nz] - ref[nz]) / ref[nz]) < 1e-9

"""BASELINE.json full sizes -- configs[1] (10 k families, 16 taxa), configs[2] (100 k families, 32 taxa,
lambda/mu) and one GPU shard of configs[3] (500 k / 8 = 62,464 families, 64 taxa) -- through size-independent
properties, plus an oracle spot check on a random sample:
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


@pytest.fixture(scope="module", params=[("cfg2", 10000), ("cfg3", 100000), ("cfg4", 62464)],
                ids=["configs1-10k", "configs2-100k", "configs3-shard-62k"])
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
    lam = np.full(t.n_nodes, cfg["lam"])
    mu = np.full(t.n_nodes, cfg["mu"])
    eng = cafe_amd.Engine(0)
    eng.set_tree(t.parent, t.left, t.right, t.branchlength)
    fr = cafe_amd.FamilySizeRange(rng.min, rng.max, rng.root_min, rng.root_max)
    eng.set_families(counts, fr)
    full = eng.get_posterior(lam, mu, prior, per_family=True)
    yield dict(eng=eng, t=t, counts=counts, rng=rng, fr=fr, prior=prior, lam=lam, mu=mu, full=full, cfg=cfg)
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


def test_oracle_spot_check_on_a_sample(problem):
    p = problem
    score, fz, ml, am, mp = p["full"]
    rs = np.random.RandomState(3)
    idx = np.sort(rs.choice(len(p["counts"]), 160, replace=False))
    so, fzo, mlo, amo, mpo = O.eval_posterior(p["t"], p["counts"][idx], p["rng"], p["lam"], p["mu"], p["prior"],
                                              nthreads=os.cpu_count() or 1)
    assert np.max(np.abs(ml[idx] - mlo) / mlo) < 1e-9
    assert np.max(np.abs(mp[idx] - mpo) / mpo) < 1e-9
    assert np.array_equal(am[idx], amo)


def test_identity_error_model_is_a_no_op(problem):
    p = problem
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

"""Pre-armed chain (option prearm=1, round 5): the next evaluation's launches wait behind a gate kernel and are released by
one store.  Same kernels, same inputs: every value must equal the ordinary path's, whatever is called in between."""
import time

import numpy as np
import pytest

from tests import _orc as O

pytestmark = pytest.mark.gpu


def _table(F=3000, prearm=1, error_model=False):
    import cafe_amd
    from cafe_amd import synth
    tree, counts, cfg = synth.make_config("cfg2", F=F)
    rng = cafe_amd.init_family_size(cfg["m"])
    eng = cafe_amd.Engine(0)
    eng.set_option("k2tune", 0)          # (the wave grid is settled from the first evaluation on)
    eng.set_option("prearm", prearm)
    tree.apply(eng)
    eng.set_families(counts, rng)
    if error_model:
        eng.set_error_model(synth.banded_error_matrix(rng.max))
    prior = O.prior_poisson(1000, rng.root_min, 8.0)
    return eng, tree, rng, prior


def _rates(tree, n, seed=1, per_node=False):
    r = np.random.default_rng(seed)
    nl = np.empty((n, tree.n_nodes))
    for i in range(n):
        nl[i] = 0.001 + 0.002 * (r.random(tree.n_nodes) if per_node and i % 3 == 0 else r.random())
    return nl, np.full_like(nl, -1.0)


@pytest.mark.parametrize("error_model", [False, True])
def test_a_loop_of_evaluations_on_pre_armed_chains_equals_the_ordinary_loop(error_model):
    eng, tree, rng, prior = _table(error_model=error_model)
    ref, _, _, _ = _table(prearm=0, error_model=error_model)
    nl, nm = _rates(tree, 40, per_node=True)     # every third set has per-node rates: another number of distinct matrices
    want = [ref.get_posterior(nl[i], nm[i], prior) for i in range(40)]
    got = [eng.get_posterior(nl[i], nm[i], prior) for i in range(40)]
    assert got == want
    st = eng.prearm_stats()
    assert st["used"] >= 5, st                      # the runs of equal shape rode on armed chains
    assert st["let_go"] >= 3, st                     # ... the shape changes let theirs go
    # per-family outputs of an evaluation that rode on a chain
    a = eng.get_posterior(nl[1], nm[1], prior, per_family=True)
    b = ref.get_posterior(nl[1], nm[1], prior, per_family=True)
    assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:]))
    eng.close()
    ref.close()


def test_other_calls_between_evaluations_and_an_expired_gate():
    eng, tree, rng, prior = _table(F=1500)
    ref, _, _, _ = _table(F=1500, prearm=0)
    nl, nm = _rates(tree, 12, seed=5)
    rows = np.random.default_rng(2).integers(0, 30, (32, (tree.n_nodes + 1) // 2)).astype(np.int32)
    lo, hi, cm = np.full(32, rng.root_min, np.int32), np.full(32, rng.root_max, np.int32), np.full(32, rng.max, np.int32)
    for i in range(12):
        assert eng.get_posterior(nl[i], nm[i], prior) == ref.get_posterior(nl[i], nm[i], prior), i
        if i % 4 == 1:      # the matrices "of the last evaluation" are this evaluation's, not a waiting chain's
            assert np.array_equal(eng.get_matrix(2), ref.get_matrix(2))
            assert np.array_equal(eng.eval_root_likelihoods(rows, lo, hi, cm), ref.eval_root_likelihoods(rows, lo, hi, cm))
        if i % 4 == 2:      # another prior: its K1 must mirror it -- no armed chain carries that
            p2 = O.prior_poisson(1000, rng.root_min, 5.0 + i)
            assert eng.get_posterior(nl[i], nm[i], p2) == ref.get_posterior(nl[i], nm[i], p2)
        if i % 4 == 3:      # the host stays away longer than the gate waits (20 ms): the chain repeats the previous
            time.sleep(0.06)    # evaluation on its own, the next call notices and launches the ordinary way
    st = eng.prearm_stats()
    assert st["used"] >= 1 and st["let_go"] >= 3, st
    eng.close()
    ref.close()


def test_a_host_kept_off_its_core_longer_than_the_gate_waits():
    # found by the round-5 soak run: the thread that waits for the score is descheduled for more than the gate's 20 ms, the
    # chain armed behind the evaluation expires, runs as a repetition of it and publishes the NEXT sequence number where
    # the evaluation's own stood.  The repetition carries the same values: the call must return them, not fail.
    eng, tree, rng, prior = _table(F=1200)
    ref, _, _, _ = _table(F=1200, prearm=0)
    nl, nm = _rates(tree, 6, seed=9)
    eng.set_option("test_stall_ms", 45)
    for i in range(6):
        a = eng.get_posterior(nl[i], nm[i], prior, per_family=True)
        b = ref.get_posterior(nl[i], nm[i], prior, per_family=True)
        assert a[0] == b[0] and a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:])), i
    st = eng.prearm_stats()
    assert st["let_go"] >= 4 and st["used"] == 0, st      # every armed chain expired behind its evaluation
    eng.close()
    ref.close()

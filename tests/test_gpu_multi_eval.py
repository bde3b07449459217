"""Several parameter sets in one pass (cafehip_eval_posterior_multi) and the batched candidate evaluation of the
searches built on it (libcommon/fminsearch.cpp:198-237): values bit-identical to single
evaluations, search trajectories identical to the sequential run."""
import os
import time

import numpy as np
import pytest

from tests import _orc as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NEWICK = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"


def _example_engine():
    import cafe_amd
    sp, ids, counts = O.load_family_table(os.path.join(GOLD, "example_data.tab"))
    t = O.PyTree(NEWICK)
    counts = O.reorder_to_tree(sp, counts, t)
    rng = O.range_from_max(int(counts.max()))
    eng = cafe_amd.Engine(0)
    eng.set_tree(t.parent, t.left, t.right, t.branchlength)
    eng.set_families(counts, cafe_amd.FamilySizeRange(rng.min, rng.max, rng.root_min, rng.root_max))
    return eng, t, rng


@pytest.mark.parametrize("with_mu", [False, True])
def test_multi_set_scores_equal_single_evaluations_bit_for_bit(with_mu):
    eng, t, rng = _example_engine()
    prior = O.prior_poisson(1000, rng.root_min, 9.442907)
    lams = [0.0017, 0.0031, 0.0005, 0.012, 0.0017, 0.2, 0.0009, 0.004]      # a repeat and a zero-matrix rate (lambda*t >= 1)
    nl = np.array([np.full(t.n_nodes, x) for x in lams])
    nm = np.array([np.full(t.n_nodes, 0.7 * x if with_mu else -1.0) for x in lams])
    singles = [eng.get_posterior(nl[i], nm[i], prior) for i in range(len(lams))]
    for n in (2, 3, 8):
        scores, fz = eng.get_posterior_multi(nl[:n], nm[:n], prior)
        for i in range(n):
            s1, z1 = singles[i]
            assert (scores[i] == s1 or (np.isinf(scores[i]) and np.isinf(s1))) and fz[i] == z1, (n, i)
    # and a single evaluation afterwards is unaffected
    assert eng.get_posterior(nl[1], nm[1], prior) == singles[1]
    eng.close()


def test_multi_set_on_a_table_of_several_workgroups_with_error_model():
    import cafe_amd
    from cafe_amd import synth
    tree, counts, cfg = synth.make_config("cfg2", F=3000)
    rng = cafe_amd.init_family_size(cfg["m"])
    eng = cafe_amd.Engine(0)
    tree.apply(eng)
    eng.set_families(counts, rng)
    eng.set_error_model(synth.banded_error_matrix(rng.max))
    prior = O.prior_poisson(1000, rng.root_min, 8.0)
    nl = np.array([np.full(tree.n_nodes, 0.002 * (1 + 0.1 * i)) for i in range(5)])
    nm = np.full_like(nl, -1.0)
    singles = [eng.get_posterior(nl[i], nm[i], prior) for i in range(5)]
    scores, fz = eng.get_posterior_multi(nl, nm, prior)
    assert [(float(scores[i]), int(fz[i])) for i in range(5)] == [(float(a), int(b)) for a, b in singles]
    eng.close()


def _search(lines, speculate):
    from cafe_amd.shell import CafeShell
    if True:
        sh = CafeShell(0, os.devnull)
        sh.set_option("speculate", 1 if speculate else 0)
        t0 = time.perf_counter()
        for l in lines:
            sh.dispatch(l)
        wall = time.perf_counter() - t0
        res = (list(sh.params), sh.score, sh.iterations, sh.evaluations, sh.trace().tolist())
        stats = sh.speculation_stats()
        secs = sh.search_seconds
        sh.close()
    return res, stats, secs, wall


@pytest.mark.parametrize("command", ["lambda -s", "lambdamu -s", "lambda -s -t ((1,1)1,(2,2)2,2)"])
def test_batched_candidates_leave_the_search_trajectory_unchanged(command):
    newick = NEWICK
    if "-t" in command:
        command = "lambda -s -t (((1,1)1,(2,2)2)2,2)"
    lines = ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + newick, command]
    seq, st0, t_seq, _ = _search(lines, False)
    spc, st1, t_spc, _ = _search(lines, True)
    assert spc == seq                                     # parameters, score, iterations, evaluations, every objective call
    assert st0 == (0, 0, 0) and st1[0] > 0 and st1[2] > 0.8 * seq[3]     # nearly every call was served from a batched pass
    print("%s: %d evaluations, sequential %.1f ms, batched %.1f ms (%d passes, %d points)"
          % (command, seq[3], 1e3 * t_seq, 1e3 * t_spc, st1[0], st1[1]))


def test_lambda_grid_batched_equals_sequential(tmp_path):
    out0, out1 = str(tmp_path / "g0.txt"), str(tmp_path / "g1.txt")
    base = ["seed 10", "load -i %s -t 1" % os.path.join(GOLD, "example_data.tab"), "tree " + NEWICK]
    _search(base + ["lambda -r 0.0005:0.0005:0.01 -o " + out0], False)
    _, st, _, _ = _search(base + ["lambda -r 0.0005:0.0005:0.01 -o " + out1], True)
    assert open(out0).read() == open(out1).read() and st[0] > 0


def test_evaluation_on_addresses_equals_the_array_wrapper():
    # bench.py's loop calls the ABI on addresses taken once (Engine.get_posterior_at): same call, same bits
    eng, t, rng = _example_engine()
    try:
        prior = np.ascontiguousarray(O.prior_poisson(1000, rng.root_min, 9.442907))
        for lam in (0.0017, 0.0031, 0.2):
            nl, nm = np.full(t.n_nodes, lam), np.full(t.n_nodes, -1.0)
            want = eng.get_posterior(nl, nm, prior)
            got = eng.get_posterior_at(eng.address_of(nl), eng.address_of(nm), eng.address_of(prior))
            assert got == want or (np.isinf(got[0]) and np.isinf(want[0]) and got[1] == want[1])
        with pytest.raises(ValueError):
            eng.address_of(np.arange(10.0)[::2])
    finally:
        eng.close()


def test_a_sequence_of_evaluations_in_one_call_equals_the_single_calls():
    # cafehip_eval_posterior_sequence: n evaluations one after the other inside the library (a `lambda -r` grid on a table
    # that fills the chip, a likelihood profile): the same calls, the same bits
    eng, t, rng = _example_engine()
    try:
        prior = np.ascontiguousarray(O.prior_poisson(1000, rng.root_min, 9.442907))
        lams = [0.0017, 0.0031, 0.2, 0.0005, 0.0017]
        nl = np.array([np.full(t.n_nodes, x) for x in lams])
        nm = np.full_like(nl, -1.0)
        singles = [eng.get_posterior(nl[i], nm[i], prior) for i in range(len(lams))]
        scores, fz = eng.get_posterior_sequence(nl, nm, prior)
        for i, (s1, z1) in enumerate(singles):
            assert (scores[i] == s1 or (np.isinf(scores[i]) and np.isinf(s1))) and fz[i] == z1, i
    finally:
        eng.close()

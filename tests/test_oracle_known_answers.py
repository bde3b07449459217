"""Pin oracle/cafe_oracle.c against the reference's own known answers.

Every expected value below is a number asserted by the reference's unit tests (file:line cited)
or printed in its golden transcripts (tests/golden/transcripts.json).  CPU only.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from tests import _orc as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TR = json.load(open(os.path.join(GOLD, "transcripts.json")))


def test_chooseln_values():
    # tests/test.cpp:790-803
    L = O.lib()
    assert L.orc_chooseln(8, 5) == pytest.approx(4.025, abs=1e-3)
    assert L.orc_chooseln(3, 2) == pytest.approx(1.098, abs=1e-3)
    assert L.orc_chooseln(6, 5) == pytest.approx(1.791, abs=1e-3)
    assert L.orc_chooseln(9, 3) == pytest.approx(4.43, abs=1e-3)
    # libcommon/mathfunc.c:224-229 edge cases
    assert L.orc_chooseln(5, 0) == 0.0
    assert L.orc_chooseln(0, 0) == 0.0
    assert L.orc_chooseln(0, 3) == -math.inf
    assert L.orc_chooseln(-1, 2) == -math.inf


def test_birthdeath_rate_with_log_alpha():
    # tests/test.cpp:805-815
    L = O.lib()
    T = L.orc_chooseln_table(60)
    assert L.orc_birthdeath_rate_with_log_alpha(40, 42, -1.37, 0.5, T, 60) == pytest.approx(0.107, abs=1e-3)
    assert L.orc_birthdeath_rate_with_log_alpha(41, 34, -1.262, 0.4, T, 60) == pytest.approx(0.006, abs=1e-3)
    assert L.orc_birthdeath_rate_with_log_alpha(5, 5, -1.193124100281034, 0.3934553412290217, T, 60) == \
        pytest.approx(0.19466, abs=1e-5)


def test_birthdeath_likelihood_with_s_c():
    # tests/test.cpp:817-823
    L = O.lib()
    T = L.orc_chooseln_table(60)
    assert L.orc_birthdeath_likelihood_with_s_c(40, 42, 0.42, 0.5, -1, T, 60) == pytest.approx(0.083, abs=1e-3)
    assert L.orc_birthdeath_likelihood_with_s_c(41, 34, 0.54, 0.4, -1, T, 60) == pytest.approx(0.023, abs=1e-3)
    # s == 0 rows, libtree/birthdeath.c:85-92
    assert L.orc_birthdeath_likelihood_with_s_c(0, 0, 1, 0.1, -1, T, 60) == 1.0
    assert L.orc_birthdeath_likelihood_with_s_c(0, 3, 1, 0.1, -1, T, 60) == 0.0


def test_compute_birthdeath_rates_lambda_mu():
    # tests/test.cpp:873-889
    m = O.birthdeath_matrix(10, 0.02, 0.01, 3)
    assert m[0, 0] == 1 and m[0, 1] == 0 and m[0, 2] == 0
    assert m[1, 0] == pytest.approx(0.086, abs=1e-3)
    assert m[1, 1] == pytest.approx(0.754, abs=1e-3)
    assert m[1, 2] == pytest.approx(0.131, abs=1e-3)
    assert m[2, 0] == pytest.approx(0.007, abs=1e-3)
    assert m[2, 1] == pytest.approx(0.131, abs=1e-3)
    assert m[2, 2] == pytest.approx(0.591, abs=1e-3)


def test_compute_birthdeath_rates_lambda_only():
    # tests/test.cpp:914-932
    m = O.birthdeath_matrix(1, 0.01, -1, 20)
    exp = {(1, 0): 0.0099, (1, 1): 0.980296, (1, 2): 0.0097059, (2, 0): 9.8e-5, (2, 1): 0.0194118,
           (2, 2): 0.961173, (3, 1): 0.000288294, (3, 2): 0.0285468}
    for (s, c), v in exp.items():
        assert m[s, c] == pytest.approx(v, abs=1e-6)


def test_no_truncation_inside_compute_but_in_cache():
    # tests/test.cpp:891-898: compute_birthdeath_rates itself uses the double branch length
    assert O.birthdeath_matrix(68.7105, 0.006335, -1, 140)[5, 5] == pytest.approx(0.19466, abs=1e-5)
    # tests/test.cpp:1015-1022: the cache key truncates, 68.7105 and 68 give the same matrix
    t = O.PyTree("(A:68.7105,B:68)")
    lam = np.full(3, 0.006335)
    mu = np.full(3, -1.0)
    ct = t.ctree()
    h = O.lib().orc_matrices_build(C.byref(ct), O.dptr(lam), O.dptr(mu), 140, 1)
    assert O.lib().orc_matrices_nkeys(h) == 1
    S = O.lib().orc_matrices_size(h)
    a = np.ctypeslib.as_array(O.lib().orc_matrices_get(h, 0), shape=(S, S))
    b = np.ctypeslib.as_array(O.lib().orc_matrices_get(h, 2), shape=(S, S))
    assert a[5, 5] == pytest.approx(0.195791, abs=1e-6)
    assert b[5, 5] == a[5, 5]
    O.lib().orc_matrices_free(h)


def test_degenerate_matrices():
    # libtree/birthdeath.c:184-225: coeff <= 0 -> zero rows; coeff == 1 -> identity; row 0 = e0
    z = O.birthdeath_matrix(100, 0.02, -1, 10)  # lambda*t = 2 -> coeff < 0
    assert z[0, 0] == 1 and np.all(z[0, 1:] == 0) and np.all(z[1:] == 0)
    i = O.birthdeath_matrix(0, 0.02, -1, 10)  # t = 0 -> alpha = 0, coeff = 1
    assert np.array_equal(i, np.eye(11))


def test_square_matrix_multiply():
    # tests/test.cpp:842-871
    L = O.lib()
    m = np.arange(1, 10, dtype=float).reshape(3, 3)
    v = np.array([7.0, 9.0, 11.0])
    out = np.zeros(3)
    L.orc_square_matrix_multiply(O.dptr(m), 3, O.dptr(v), 0, 2, 0, 2, O.dptr(out))
    assert list(out) == [58, 139, 220]
    big = np.zeros((8, 8))
    big[3:6, 3:6] = m
    L.orc_square_matrix_multiply(O.dptr(big), 8, O.dptr(v), 3, 5, 3, 5, O.dptr(out))
    assert list(out) == [58, 139, 220]


def test_compute_tree_likelihood_small_tree():
    # tests/test.cpp:441-474: ((A:1,B:1):1,(C:1,D:1):1), lambda .01, leaves 5,3,2,4, ranges 0..7
    t = O.PyTree("((A:1,B:1):1,(C:1,D:1):1)")
    rng = O.make_range(0, 7, 0, 7)
    lam = np.full(t.n_nodes, 0.01)
    mu = np.full(t.n_nodes, -1.0)
    ct = t.ctree()
    L = O.lib()
    h = L.orc_matrices_build(C.byref(ct), O.dptr(lam), O.dptr(mu), 7, 1)
    fs = np.full(t.n_nodes, -1, np.int32)
    fs[[0, 2, 4, 6]] = [5, 3, 2, 4]
    sof = 8
    Lbuf = np.zeros(t.n_nodes * sof)
    L.orc_compute_tree_likelihoods(C.byref(ct), C.byref(rng), h, O.iptr(fs), None, 0, None, O.dptr(Lbuf), sof)
    lk = Lbuf[t.root * sof:(t.root + 1) * sof]
    assert lk[0] == pytest.approx(0, abs=1e-10)
    assert lk[1] == pytest.approx(1.42138e-13, abs=1e-13)
    assert lk[2] == pytest.approx(2.87501e-09, abs=1e-13)
    assert lk[3] == pytest.approx(4.11903e-07, abs=1e-7)
    assert lk[4] == pytest.approx(6.73808e-07, abs=1e-7)
    L.orc_matrices_free(h)


def test_prior_poisson():
    # tests/test.cpp:656-670: shift 1, lambda 5.75
    p = O.prior_poisson(6, 1, 5.75)
    for got, exp in zip(p, [0.00318278, 0.018301, 0.0526153, 0.100846, 0.144966, 0.166711]):
        assert got == pytest.approx(exp, abs=1e-6)


def test_find_poisson_lambda():
    # tests/lambda_tests.cpp:667-696: counts {6,11,3,7} x 4 species... -> 5.75
    counts = np.array([[6, 11, 3, 7]] * 4, np.int32)
    it = C.c_int()
    sc = C.c_double()
    lam = O.lib().orc_find_poisson_lambda(4, 4, O.iptr(counts), 0.5, C.byref(it), C.byref(sc))
    assert lam == pytest.approx(5.75, abs=1e-3)


def test_pvalue():
    # tests/test.cpp:1627-1633
    cd = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9])
    assert O.lib().orc_pvalue(0.35, O.dptr(cd), 9) == pytest.approx(3.0 / 9.0)


def test_get_posterior_real_matrices():
    # tests/lambda_tests.cpp:698-742: (A:1,B:1), lambda 0.27290862102823, 4 families,
    # Poisson(2) prior, range 0..149 / 1..124 -> -18.0085 +- .1
    t = O.PyTree("(A:1,B:1)")
    counts = np.array([[1, 2], [2, 1], [3, 6], [6, 3]], np.int32)
    rng = O.make_range(0, 149, 1, 124)
    prior = O.prior_poisson(1000, 1, 2.0)
    lam = np.full(t.n_nodes, 0.27290862102823)
    mu = np.full(t.n_nodes, -1.0)
    score, fz, *_ = O.eval_posterior(t, counts, rng, lam, mu, prior)
    assert score == pytest.approx(-18.0085, abs=0.1)


def test_survey_8c_example_pins():
    # outputs of the reference run in the build container, SURVEY.md section 8(c), to 17 digits
    g = TR["survey_8c_example"]
    t = O.PyTree(g["newick"])
    sp, ids, counts = O.load_family_table(os.path.join(GOLD, "example_data.tab"))
    counts = O.reorder_to_tree(sp, counts, t)
    rng = O.range_from_max(int(counts.max()))
    assert (rng.max + 1, rng.root_max - rng.root_min + 1) == (g["S"], g["rfsize"])
    prior = O.prior_poisson(1000, rng.root_min, g["poisson_lambda"])
    lam = np.full(t.n_nodes, g["lambda"])
    mu = np.full(t.n_nodes, -1.0)
    score, fz, ml, am, mp = O.eval_posterior(t, counts, rng, lam, mu, prior)
    assert fz == -1
    for fid, (exp_ml, exp_lp) in g["families"].items():
        i = ids.index(fid)
        assert ml[i] == pytest.approx(exp_ml, rel=1e-15)
        assert math.log(mp[i]) == pytest.approx(exp_lp, rel=1e-15)
    M = max(rng.max, rng.root_max)
    assert O.birthdeath_matrix(6, g["lambda"], -1, M)[5, 5] == pytest.approx(g["P_bl6_5_5"], rel=1e-15)


def _load(name):
    g = TR[name]
    t = O.PyTree(g["newick"])
    fn = {"test1": "test1_families.txt.gz", "test2": "test2_families.txt"}[name]
    sp, ids, counts = O.load_family_table(os.path.join(GOLD, fn), max_size=g["max_size"])
    counts = O.reorder_to_tree(sp, counts, t)
    assert len(ids) == g["n_families"]
    rng = O.range_from_max(int(counts.max()))
    assert [rng.root_min, rng.root_max] == g["root_range"]
    assert [rng.min, rng.max] == g["family_range"]
    return g, t, counts, rng


def test_transcript_test2_all_pairs():
    # tests/integration/test2.t: every "Lambda : x & Score: y" line of the search, 4 families
    g, t, counts, rng = _load("test2")
    lp = O.lib().orc_find_poisson_lambda(counts.shape[0], counts.shape[1], O.iptr(counts), 0.5, None, None)
    assert lp == pytest.approx(g["poisson_lambda"], abs=2e-6)
    prior = O.prior_poisson(1000, rng.root_min, g["poisson_lambda"])
    mu = np.full(t.n_nodes, -1.0)
    for lam_v, exp in g["lambda_score"]:
        if lam_v < 0:
            assert exp == -math.inf  # cafe/lambda.cpp:733-741
            continue
        score, fz, *_ = O.eval_posterior(t, counts, rng, np.full(t.n_nodes, lam_v), mu, prior)
        assert score == pytest.approx(exp, abs=2e-6)


def test_transcript_test1_pairs():
    # tests/integration/test1.t: 14,787 families x 20 taxa; printed Poisson lambda has 6 decimals,
    # which bounds the agreement at ~3e-3 absolute on a score of 4.7e5 (6e-9 relative)
    g, t, counts, rng = _load("test1")
    it = C.c_int()
    sc = C.c_double()
    lp = O.lib().orc_find_poisson_lambda(counts.shape[0], counts.shape[1], O.iptr(counts), 0.5, C.byref(it), C.byref(sc))
    assert lp == pytest.approx(g["poisson_lambda"], abs=5e-6)
    assert sc.value == pytest.approx(g["poisson_score"], abs=1e-3)
    prior = O.prior_poisson(1000, rng.root_min, g["poisson_lambda"])
    mu = np.full(t.n_nodes, -1.0)
    pairs = g["lambda_score"]
    for lam_v, exp in [pairs[0], pairs[6], pairs[-1]]:
        score, fz, *_ = O.eval_posterior(t, counts, rng, np.full(t.n_nodes, lam_v), mu, prior,
                                         nthreads=os.cpu_count() or 1)
        assert fz == -1
        assert score == pytest.approx(exp, abs=5e-3)


def test_ref_duplicates_and_zero_family():
    # cafe/cafe_family.c:9-34 and cafe/lambda.cpp:715-720
    t = O.PyTree("((A:10,B:10):5,C:15)")
    counts = np.array([[1, 2, 3], [4, 4, 4], [1, 2, 3], [0, 0, 0], [4, 4, 4]], np.int32)
    ref = np.zeros(5, np.int32)
    O.lib().orc_family_check_the_pattern(5, 3, O.iptr(counts), O.iptr(ref))
    assert list(ref) == [0, 1, 0, 3, 1]
    rng = O.range_from_max(4)
    prior = O.prior_poisson(1000, 1, 2.0)
    lam = np.full(t.n_nodes, 0.01)
    mu = np.full(t.n_nodes, -1.0)
    s1, fz1, ml1, am1, mp1 = O.eval_posterior(t, counts, rng, lam, mu, prior, ref=ref)
    s2, fz2, ml2, am2, mp2 = O.eval_posterior(t, counts, rng, lam, mu, prior)
    assert s1 == s2 and np.array_equal(mp1, mp2)
    # lambda * t >= 1 on an edge -> zero matrix -> likelihood 0 -> score -inf, first family reported
    lam_big = np.full(t.n_nodes, 0.2)
    s3, fz3, ml3, *_ = O.eval_posterior(t, counts, rng, lam_big, mu, prior)
    assert s3 == -math.inf and fz3 == 0 and ml3[0] == 0.0


def test_report_golden_test2_cafe():
    # tests/integration/test2.cafe (golden report of test2.sh, seed 10, -p 0.05): family-wide p-values,
    # Viterbi ancestral sizes and branch p-values -- pins the MC null (libc rand() stream), pvalue(),
    # the max-product pass and viterbi_sum_probabilities.
    g = TR["test2"]
    t = O.PyTree(g["newick"])
    sp, ids, counts = O.load_family_table(os.path.join(GOLD, "test2_families.txt"), max_size=g["max_size"])
    counts = O.reorder_to_tree(sp, counts, t)
    rng = O.range_from_max(int(counts.max()))
    rep, cd = O.report_with_oracle(t, counts, rng, g["search_result"]["lambda"], pvalue_cut=0.05)
    gold = O.parse_cafe_report(os.path.join(GOLD, "test2.cafe"))
    assert len(gold) == 4
    # Newick print order of (((Chimp,Human),(Mouse,Rat)),Dog): ids 0,2,1,4,6,5,3,8,7
    order = [0, 2, 1, 4, 6, 5, 3, 8, 7]
    for fid, (maxp, sizes, bp) in zip(ids, rep):
        gs, gp, gpairs = gold[fid]
        assert [int(sizes[i]) for i in order] == gs
        assert float("%g" % maxp) == gp
        for j, pair in enumerate(gpairs):
            if pair is None:
                assert bp is None
            else:
                assert float("%g" % bp[2 * j]) == pair[0] and float("%g" % bp[2 * j + 1]) == pair[1]


def test_transcript_test3_two_class_search_lines():
    # tests/integration/test3.t, first search (lambda -s -t (((2,2)1,(1,1)1)1,1) on the example table): every
    # printed "Lambda : l1,l2 & Score" line of the 117-iteration two-class search against the oracle.
    # Class 2 = chimp and human (nodes 0 and 2 of the in-order numbering), class 1 = everything else.
    import gzip
    import json
    ev = json.load(gzip.open(os.path.join(GOLD, "test3_transcript.json.gz"), "rt"))["events"]
    newick = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"
    sp, ids, counts = O.load_family_table(os.path.join(GOLD, "example_data.tab"))
    t = O.PyTree(newick)
    counts = O.reorder_to_tree(sp, counts, t)
    rng = O.range_from_max(int(counts.max()))
    first_poisson = next(e for e in ev if e[0] == "poisson")
    prior = O.prior_poisson(1000, rng.root_min, first_poisson[1])
    cls2 = np.zeros(t.n_nodes, bool)
    cls2[[0, 2]] = True
    mu = np.full(t.n_nodes, -1.0)
    n = 0
    for e in ev:
        if e[0] == "result":
            break
        if e[0] != "eval":
            continue
        (l1, l2), exp = e[1], e[2]
        if l1 < 0 or l2 < 0:
            assert exp == -math.inf
            continue
        if abs(l1 * 93 - 1) < 1e-9 or abs(l2 * 6 - 1) < 1e-9:
            continue  # on a lambda*t = 1 cliff the 14 printed decimals do not decide the side
        score, fz, *_ = O.eval_posterior(t, counts, rng, np.where(cls2, l2, l1), mu, prior)
        if math.isinf(exp):
            assert score == exp
        else:
            # the printed prior lambda (6 decimals) bounds the agreement
            assert score == pytest.approx(exp, abs=2e-4)
        n += 1
    assert n > 150

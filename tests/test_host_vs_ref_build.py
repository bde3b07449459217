"""The PRODUCT's host routines whose exact behaviour steers results -- the rank of a likelihood in a sorted Monte-Carlo
null (cafe_host.cpp pvalue_rank) and the Nelder-Mead of the lambda searches (cafe_host.cpp FMinSearch) -- bit for bit
against the reference's own objects: oracle/_ref/libcaferef.so = libcommon/mathfunc.c + libcommon/fminsearch.cpp compiled
straight from /root/reference by oracle/Makefile.  (tests/test_oracle_vs_ref_build.py pins the ORACLE the same way; this
file pins the library that ships.)  CPU only; skipped when neither the prebuilt library nor the reference tree exists."""
import ctypes as C
import math

import numpy as np
import pytest

from tests import _orc as O
from tests.test_oracle_vs_ref_build import _ref


def _host():
    from cafe_amd import _lib
    return _lib.load()


def test_pvalue_rank_bitwise_against_the_reference():
    R, H = _ref(), _host()
    rs = np.random.RandomState(11)
    n_checked = 0
    for trial in range(400):
        n = int(rs.randint(1, 60))
        digits = int(rs.randint(0, 3))                     # 0 digits: long runs of ties
        cd = np.sort(np.round(rs.rand(n), digits))
        probes = list(cd[rs.randint(0, n, 4)]) + [0.0, 1.0, 0.05, 0.35, 0.5, 2.0, -1.0, float(cd[0]), float(cd[-1])]
        for v in probes:
            assert H.cafehost_pvalue_selftest(float(v), O.dptr(cd), n) == R.pvalue(float(v), O.dptr(cd), n), (trial, v, cd)
            n_checked += 1
    # the shape the report uses: 1000 sorted likelihoods, values between them and on them
    cd = np.sort(np.exp(-30 * rs.rand(1000)))
    cd[100:110] = cd[100]
    for v in list(cd[::37]) + [cd[100], cd[105], 0.0, 1.0, 1e-300, float(cd[0]) / 2]:
        assert H.cafehost_pvalue_selftest(float(v), O.dptr(cd), 1000) == R.pvalue(float(v), O.dptr(cd), 1000)
    assert n_checked > 4000


def _run_both(f, x0, tol=1e-6):
    R, H = _ref(), _host()
    N = len(x0)
    trace = {"ref": [], "host": []}

    def mk(tag):
        def cb(xp, _):
            x = [xp[i] for i in range(N)]
            trace[tag].append(tuple(x))
            return f(x)
        return O.MATH_FUNC(cb)

    x0a = np.array(x0, float)
    xr, xh = np.zeros(N), np.zeros(N)
    fr, fh = C.c_double(), C.c_double()
    br, bh = C.c_int(), C.c_int()
    it_r = R.ref_fminsearch(mk("ref"), N, None, O.dptr(x0a.copy()), tol, tol, O.dptr(xr), C.byref(fr), C.byref(br))
    cb = mk("host")
    it_h = H.cafehost_fminsearch_selftest(C.cast(cb, C.c_void_p), N, None, O.dptr(x0a.copy()), tol, tol, O.dptr(xh), C.byref(fh), C.byref(bh))
    return (it_r, xr, fr.value, br.value, trace["ref"]), (it_h, xh, fh.value, bh.value, trace["host"])


def _same(ref, host):
    assert ref[0] == host[0], (ref[0], host[0])            # iterations
    assert ref[4] == host[4]                               # every point asked for, in order, bit for bit
    assert np.array_equal(ref[1], host[1]) and ref[2] == host[2] and ref[3] == host[3]


def test_nelder_mead_same_trajectory_1d_with_inf_region():
    # the lambda objective returns +inf for a negative rate (cafe/lambda.cpp:733-741)
    _same(*_run_both(lambda x: math.inf if x[0] < 0 else (x[0] - 0.0123) ** 2 + 3.0, [0.4]))


def test_nelder_mead_same_trajectory_3d():
    _same(*_run_both(lambda x: (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2 + (x[2] - 0.5) ** 4, [-1.2, 1.0, 0.0]))


def test_nelder_mead_isinf_start_simplex_rule():
    # libcommon/fminsearch.cpp:158-167: after an infinite vertex value the next vertex uses delta*100
    _same(*_run_both(lambda x: math.inf if (x[0] > 1.04 or x[1] < 0) else (x[0] - 0.3) ** 2 + (x[1] - 0.2) ** 2, [1.0, 0.5]))


def test_nelder_mead_ties_and_plateaus_keep_the_reference_order():
    # equal values at several vertices: the unstable sort's tie order decides which vertex is replaced next
    _same(*_run_both(lambda x: float(round(abs(x[0] - 0.31) + abs(x[1] - 0.17) + abs(x[2] + 0.4), 1)), [0.9, 0.9, 0.9], tol=1e-3))
    _same(*_run_both(lambda x: 1.0 if abs(x[0]) + abs(x[1]) > 0.5 else 0.0, [1.0, 1.0], tol=1e-4))
    _same(*_run_both(lambda x: max(abs(x[0] - 1), abs(x[1] + 2), abs(x[2]), abs(x[3] - 0.5)), [0.0, 0.0, 0.1, 0.2]))


def test_nelder_mead_nan_scores_take_the_same_branches():
    # the k-cluster objective can return NaN (cafe/cafe_main.c:204, 0/0 membership): every comparison with it is false
    f = lambda x: math.nan if 0.45 < x[0] < 0.55 else (x[0] - 0.2) ** 2 + (x[1] - 0.1) ** 2
    _same(*_run_both(f, [0.5, 0.5]))


def test_poisson_prior_fit_with_lookahead_is_the_plain_fit_bit_for_bit():
    # find_poisson_lambda (cafe/lambda.cpp:771-838): the objective is one long chain of dependent additions over every
    # non-zero leaf count; the look-ahead evaluates the points the 1-D Nelder-Mead may ask for several per sweep.  Fitted
    # lambda, score and iteration count must be the SAME BITS as one sweep per call -- on small tables (one core runs the
    # chains) and large ones (chains dealt to threads) -- with far fewer sweeps; and the plain fit must be the reference's
    # own fminsearch on the reference's own poisspdf (oracle/_ref).
    import time
    R, H = _ref(), _host()
    rs = np.random.RandomState(5)
    for n, lam_true, start in ((300, 3.0, 0.37), (5000, 9.4, 0.0123), (60000, 1.3, 0.9), (1200000, 7.7, 0.5557)):
        xs = np.ascontiguousarray(rs.poisson(lam_true, n).astype(np.int32))
        if n == 5000:
            xs[17] = 400        # poisspdf underflows: log(0) = -inf terms, the fit must walk through them identically
        if n == 60000:
            xs[123] = 70000     # one size beyond 16 bits: the chains fall back to 32-bit sizes, same values
        out = {}
        for la in (0, 1):
            lam, sc, it, ps = C.c_double(), C.c_double(), C.c_int(), C.c_long()
            t0 = time.time()
            assert H.cafehost_poisson_fit_selftest(xs.ctypes.data_as(C.POINTER(C.c_int32)), n, start, la, C.byref(lam), C.byref(sc),
                                                   C.byref(it), C.byref(ps)) == 0
            out[la] = (lam.value, sc.value, it.value, ps.value, time.time() - t0)
        assert out[0][:3] == out[1][:3], (n, out)
        if n >= 500000:
            assert out[1][3] < 0.75 * out[0][3], (n, out)   # sweeps over the table: at most three quarters, typically a third
        else:
            assert out[1][3] == out[0][3]                   # small tables: a sweep is microseconds, the plain fit is the fast one
        if n <= 5000:
            # ... and the plain fit is the reference's: its fminsearch, its poisspdf, the same sum
            def cb(xp, _):
                s = 0.0
                for x in xs:
                    ll = R.poisspdf(int(x), xp[0])
                    s += math.log(ll) if ll > 0 else -math.inf
                return -s
            xr, fr, br = np.zeros(1), C.c_double(), C.c_int()
            it_r = R.ref_fminsearch(O.MATH_FUNC(cb), 1, None, O.dptr(np.array([start])), 1e-6, 1e-6, O.dptr(xr), C.byref(fr), C.byref(br))
            assert (xr[0], fr.value, it_r) == out[0][:3], (n, xr[0], fr.value, it_r, out[0])
        print("poisson fit n=%d: plain %d sweeps %.3f s, look-ahead %d sweeps %.3f s" % (n, out[0][3], out[0][4], out[1][3], out[1][4]))

"""Pin the oracle bit-for-bit against the directly-compilable subset of the reference
(oracle/_ref/libcaferef.so: libcommon/mathfunc.c, libtree/chooseln_cache.c,
libcommon/fminsearch.cpp built from /root/reference by oracle/Makefile).  CPU only; skipped
when neither the prebuilt library nor the reference tree is present."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from tests import _orc as O

REF_SO = os.path.join(O.ORACLE_DIR, "_ref", "libcaferef.so")


def _ref():
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/libcommon"):
        subprocess.check_call(["make", "-C", O.ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built and no reference tree available")
    R = C.CDLL(REF_SO)
    R.gammaln.restype = C.c_double
    R.gammaln.argtypes = [C.c_double]
    R.chooseln.restype = C.c_double
    R.chooseln.argtypes = [C.c_double, C.c_double]
    R.poisspdf.restype = C.c_double
    R.poisspdf.argtypes = [C.c_int, C.c_double]
    R.pvalue.restype = C.c_double
    R.pvalue.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_int]
    R.__maxidx.restype = C.c_int
    R.__maxidx.argtypes = [C.POINTER(C.c_double), C.c_int]
    R.ref_chooseln_table.restype = None
    R.ref_chooseln_table.argtypes = [C.c_int, C.POINTER(C.c_double)]
    R.ref_fminsearch.restype = C.c_int
    R.ref_fminsearch.argtypes = [O.MATH_FUNC, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    return R


def test_gammaln_chooseln_poisspdf_bitwise():
    R, L = _ref(), O.lib()
    for a in [0.5, 1.0, 2.0, 3.5, 10.0, 151.0, 300.0, 999.0]:
        assert R.gammaln(a) == L.orc_gammaln(a)
    for n in range(0, 40):
        for r in range(0, n + 1):
            assert R.chooseln(n, r) == L.orc_chooseln(n, r)
    for x in range(0, 60):
        for lam in (0.3, 1.7, 9.442907):
            assert R.poisspdf(x, lam) == L.orc_poisspdf(x, lam)


@pytest.mark.parametrize("size", [7, 60, 150])
def test_chooseln_cache_table_bitwise(size):
    R, L = _ref(), O.lib()
    ref = np.zeros((2 * size, size + 1))
    R.ref_chooseln_table(size, O.dptr(ref))
    mine = np.ctypeslib.as_array(L.orc_chooseln_table(size), shape=(2 * size, size + 1))
    assert np.array_equal(ref, mine, equal_nan=True)


def test_pvalue_and_maxidx_bitwise():
    R, L = _ref(), O.lib()
    rs = np.random.RandomState(3)
    for _ in range(50):
        n = int(rs.randint(1, 40))
        cd = np.sort(np.round(rs.rand(n), 1))  # ties on purpose
        for v in list(cd[:3]) + [0.05, 0.35, 2.0, -1.0]:
            assert R.pvalue(float(v), O.dptr(cd), n) == L.orc_pvalue(float(v), O.dptr(cd), n)
        d = np.round(rs.rand(n), 1)
        assert R.__maxidx(O.dptr(d), n) == L.orc_maxidx(O.dptr(d), n)


def _run_both(f, x0, tol=1e-6):
    R, L = _ref(), O.lib()
    N = len(x0)
    trace = {"ref": [], "orc": []}

    def mk(tag):
        def cb(xp, _):
            x = [xp[i] for i in range(N)]
            trace[tag].append(tuple(x))
            return f(x)
        return O.MATH_FUNC(cb)

    x0a = np.array(x0, float)
    xr = np.zeros(N)
    fr = C.c_double()
    br = C.c_int()
    it_r = R.ref_fminsearch(mk("ref"), N, None, O.dptr(x0a.copy()), tol, tol, O.dptr(xr), C.byref(fr), C.byref(br))
    xo = np.zeros(N)
    fo = C.c_double()
    bo = C.c_int()
    it_o = L.orc_fminsearch(mk("orc"), N, None, O.dptr(x0a.copy()), tol, tol, 10000, O.dptr(xo), C.byref(fo), C.byref(bo))
    return (it_r, xr, fr.value, trace["ref"]), (it_o, xo, fo.value, trace["orc"])


def test_fminsearch_same_trajectory_1d_with_inf_region():
    # objective returns +inf for x < 0 like the lambda objective (cafe/lambda.cpp:733-741)
    f = lambda x: math.inf if x[0] < 0 else (x[0] - 0.0123) ** 2 + 3.0
    ref, orc = _run_both(f, [0.4])
    assert ref[0] == orc[0] and ref[3] == orc[3]
    assert np.array_equal(ref[1], orc[1]) and ref[2] == orc[2]


def test_fminsearch_same_trajectory_3d():
    f = lambda x: (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2 + (x[2] - 0.5) ** 4
    ref, orc = _run_both(f, [-1.2, 1.0, 0.0])
    assert ref[0] == orc[0] and ref[3] == orc[3]
    assert np.array_equal(ref[1], orc[1]) and ref[2] == orc[2]


def test_fminsearch_isinf_start_simplex_rule():
    # libcommon/fminsearch.cpp:158-167: after an infinite vertex value the next vertex uses delta*100
    f = lambda x: math.inf if (x[0] > 1.04 or x[1] < 0) else (x[0] - 0.3) ** 2 + (x[1] - 0.2) ** 2
    ref, orc = _run_both(f, [1.0, 0.5])
    assert ref[3][:3] == orc[3][:3]
    assert ref[0] == orc[0] and ref[3] == orc[3]

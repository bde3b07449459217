"""FMinSearch::lookahead (round 5): before each objective call the optimiser announces every point the call AFTER it may ask
for.  Host only: the announcements must leave the trajectory alone, and must really contain the next point -- bit for bit,
since the device store of matrices built ahead is keyed by exact comparison of the rates."""
import ctypes as C
import math

import numpy as np

from tests import _orc as O


def _run(f, x0, tol=1e-6):
    from cafe_amd import _lib
    H = _lib.load()
    N = len(x0)
    traces = {"plain": [], "look": []}

    def mk(tag):
        def cb(xp, _):
            x = [xp[i] for i in range(N)]
            traces[tag].append(tuple(x))
            return f(x)
        return O.MATH_FUNC(cb)

    x0a = np.array(x0, float)
    xa, xb = np.zeros(N), np.zeros(N)
    fa, fb = C.c_double(), C.c_double()
    bm = C.c_int()
    cb1 = mk("plain")
    it_a = H.cafehost_fminsearch_selftest(C.cast(cb1, C.c_void_p), N, None, O.dptr(x0a.copy()), tol, tol, O.dptr(xa), C.byref(fa), C.byref(bm))
    out = (C.c_long * 4)()
    cb2 = mk("look")
    it_b = H.cafehost_lookahead_selftest(C.cast(cb2, C.c_void_p), N, None, O.dptr(x0a.copy()), tol, tol, O.dptr(xb), C.byref(fb), out)
    assert it_a == it_b and traces["plain"] == traces["look"] and np.array_equal(xa, xb) and fa.value == fb.value
    return {"evaluations": out[0], "covered": out[1], "announcements": out[2], "points": out[3]}


def test_one_parameter_every_evaluation_but_the_first_is_foreseen():
    # the lambda search: the objective is +inf for a negative rate (cafe/lambda.cpp:733-741)
    st = _run(lambda x: math.inf if x[0] < 0 else (x[0] - 0.0123) ** 2 + 3.0, [0.4])
    assert st["evaluations"] > 30 and st["covered"] >= st["evaluations"] - 3, st
    st = _run(lambda x: abs(x[0] - 0.002) ** 1.5, [0.01])
    assert st["covered"] >= st["evaluations"] - 3, st


def test_more_parameters_the_evaluations_behind_a_reflection_an_expansion_or_a_contraction_are_foreseen():
    # (not the ones behind the initial simplex or a shrink: their order depends on values to come)
    st = _run(lambda x: (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2 + (x[2] - 0.5) ** 4, [-1.2, 1.0, 0.0])
    assert st["evaluations"] > 100 and st["covered"] >= 0.9 * st["evaluations"], st
    assert st["points"] <= 8 * st["announcements"]          # what one prefetch call takes
    st = _run(lambda x: (x[0] - 0.3) ** 2 + (x[1] - 0.2) ** 2, [1.0, 0.5])
    assert st["covered"] >= 0.85 * st["evaluations"], st


def test_plateaus_ties_and_nan_leave_the_trajectory_alone():
    # equal values at several vertices, NaN scores: whatever is announced, the points asked for are the plain loop's
    _run(lambda x: float(round(abs(x[0] - 0.31) + abs(x[1] - 0.17) + abs(x[2] + 0.4), 1)), [0.9, 0.9, 0.9], tol=1e-3)
    _run(lambda x: 1.0 if abs(x[0]) + abs(x[1]) > 0.5 else 0.0, [1.0, 1.0], tol=1e-4)
    _run(lambda x: math.nan if 0.45 < x[0] < 0.55 else (x[0] - 0.2) ** 2 + (x[1] - 0.1) ** 2, [0.5, 0.5])

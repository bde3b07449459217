"""The lambda != mu ("alpha/beta") branch of the transition matrices -- all of BASELINE configs[2]'s arithmetic --
is pinned by reference-held data only to 1e-3 (the 3x3 matrix of tests/test.cpp:873-889, checked in
test_oracle_known_answers.py); the reference has no lambdamu transcript.  oracle/alpha_beta_pin.c closes the gap
with two independent quad-precision evaluations at the configs[2] extents (M = 250):
  A  the closed form of libtree/birthdeath.c:34-50 with the reference's own ln C values, everything else in
     __float128: the oracle may differ only by the double rounding of its operation sequence (<= 1e-12);
  B  the s-fold convolution of the one-individual offspring distribution (no binomials, no log-gamma, alpha and
     beta recomputed from the rates in quad precision): the oracle differs by the reference's Lanczos ln-Gamma
     error, ~1e-10 relative, which it has to reproduce (its ln C table is bitwise the reference's,
     test_oracle_vs_ref_build.py)."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")

CASES = [
    (250, 58, 0.002, 0.0015),     # configs[2]'s true rates on a mid branch
    (250, 1, 0.002, 0.0015),      # shortest branch: alpha, beta ~ 1e-3
    (250, 100, 0.0021, 0.0014),   # root-to-tip height
    (250, 100, 0.0015, 0.002),    # mu > lambda
    (250, 17, 0.002, 0.002),      # lambda == mu >= 0 still takes the alpha/beta sum (libtree/birthdeath.c:277-280)
    (150, 96, 0.0021, 0.0001),    # strongly asymmetric rates
]


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-s", "-C", ORACLE, "liboracle.so", "alpha_beta_pin"])
    return os.path.join(ORACLE, "alpha_beta_pin")


def test_alpha_beta_matrices_against_two_independent_quad_precision_evaluations(tool):
    def run(case):
        out = subprocess.run([tool] + [repr(x) for x in case], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        m = re.match(r"A (\S+) B (\S+) rows (\d+)", out.stdout)
        assert m, out.stdout
        return float(m.group(1)), float(m.group(2)), int(m.group(3))

    with ThreadPoolExecutor(len(CASES)) as ex:
        results = list(ex.map(run, CASES))
    for case, (a, b, rows) in zip(CASES, results):
        assert rows > 20000, case
        assert a <= 1e-12, (case, a)     # measured 0.9e-13 .. 2.1e-13
        assert b <= 5e-10, (case, b)     # measured 0.5e-10 .. 1.1e-10: the reference's Lanczos ln-Gamma

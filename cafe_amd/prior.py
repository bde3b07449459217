"""Root-size prior of the lambda search (host side, once per `lambda` command).

cafe_set_prior_rfsize_poisson_lambda, cafe/lambda.cpp:841-852:
    prior[i] = poisspdf(root_min - 1 + i, lambda_p) = exp(x log(lambda_p) - lnGamma(x+1) - lambda_p)
with the reference's Lanczos lnGamma (libcommon/mathfunc.c:112-119, :352-355).
"""
import math

import numpy as np

_Q = (1.000000000190015, 76.18009172947146, -86.50532032941677, 24.01409824083091, -1.231739572450155,
      1.208650973866179e-3, -5.395239384953e-6)
_SQRT_2PI = 2.5066282746310002416123552393401042


def gammaln(a):
    p = _Q[0]
    for n in range(1, 7):
        p += _Q[n] / (a + n)
    return (a + 0.5) * math.log(a + 5.5) - (a + 5.5) + math.log(_SQRT_2PI * p / a)


def poisspdf(x, lam):
    return math.exp(x * math.log(lam) - gammaln(x + 1) - lam)


def prior_rfsize_poisson(root_min, lam, n=1000):
    return np.array([poisspdf(root_min - 1 + i, lam) for i in range(n)])


def poisson_lambda_mle(counts):
    """Closed-form optimum of the objective find_poisson_lambda minimises (cafe/lambda.cpp:771-838):
    the Poisson MLE of (count - 1) over positive counts is their mean.  The reference reaches it by
    Nelder-Mead from a random start and stops within its 1e-6 tolerance of this value."""
    c = np.asarray(counts)
    pos = c[c > 0]
    return float((pos - 1).mean())

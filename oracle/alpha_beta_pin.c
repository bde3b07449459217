/* alpha_beta_pin.c -- TEST INFRASTRUCTURE ONLY (oracle/): an independent evaluation of the lambda != mu
 * ("alpha/beta") transition matrices that pins oracle/cafe_oracle.c's restatement of
 * libtree/birthdeath.c:34-50 (the sum) and :255-262 (alpha, beta, coeff).
 *
 * The reference holds no transcript and only one 1e-3 vector for this branch
 * (tests/test.cpp:873-889), so the oracle's arithmetic there is checked two ways, both in __float128:
 *
 *  (A) the same closed form, term by term, with the ln C values the reference would read (the oracle's
 *      table, bitwise equal to the reference's chooseln_cache: tests/test_oracle_vs_ref_build.py) but every
 *      product, exponential and the sum in quad precision.  Differences from the oracle are then only
 *      the double rounding of its own operation sequence: expected <~ 1e-12 relative.
 *
 *  (B) a route that shares NOTHING with the closed form: one individual leaves c descendants with
 *      P1(0) = alpha, P1(c) = (1-alpha)(1-beta) beta^(c-1); s individuals are the s-fold convolution.
 *      No binomials, no log-gamma.  The oracle differs from it by the reference's own Lanczos ln-Gamma
 *      approximation (libcommon/mathfunc.c:112-119, ~1e-10 relative on a binomial), which the oracle must
 *      reproduce, and by the truncation of the convolution at M (none: entries c <= M of a convolution only
 *      need entries <= M).
 *
 * Usage: alpha_beta_pin M branchlength lambda mu   -> prints "A <max rel> B <max rel> rows <n compared>"
 * Build: see oracle/Makefile (links liboracle.so and libquadmath). */
#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>

#include "cafe_oracle.h"

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s M branchlength lambda mu\n", argv[0]);
        return 2;
    }
    const int M = atoi(argv[1]);
    const double t = atof(argv[2]), lambda = atof(argv[3]), mu = atof(argv[4]);
    const int S = M + 1;

    /* the oracle's matrix */
    double *P = (double *)malloc(sizeof(double) * (size_t)S * S);
    orc_compute_birthdeath_rates(t, lambda, mu, M, P);

    /* alpha, beta, coeff exactly as libtree/birthdeath.c:255-262 forms them (double) */
    double alpha, beta, coeff;
    if (mu < 0 || lambda == mu) {
        alpha = lambda * t / (1 + lambda * t);
        beta = alpha;
        coeff = 1 - 2 * alpha;
    } else {
        double e_diff = exp((lambda - mu) * t);
        double numerator = e_diff - 1;
        double denominator = lambda * (e_diff)-mu;
        alpha = (mu * numerator) / denominator;
        beta = (lambda * numerator) / denominator;
        coeff = 1 - alpha - beta;
    }
    if (!(coeff > 0) || coeff == 1) {
        printf("degenerate coeff %g: nothing to compare\n", coeff);
        return 0;
    }

    /* (A) closed form in quad precision with the reference's ln C values */
    double *lnc = orc_chooseln_table(M);
    const int ld = M + 1;
    const __float128 la = logq((__float128)alpha), lb = logq((__float128)beta), lc = logq((__float128)coeff);
    double worstA = 0;
    long n_cmp = 0;
    for (int s = 1; s <= M; s++) {
        for (int c = 0; c <= M; c++) {
            const int m = s < c ? s : c;
            __float128 p = 0;
            for (int j = 0; j <= m; j++) {
                const __float128 tq = (__float128)lnc[(size_t)s * ld + j] + (__float128)lnc[(size_t)(s + c - 1 - j) * ld + (s - 1)] +
                                      (s - j) * la + (c - j) * lb + j * lc;
                p += expq(tq);
            }
            if (p > 1) p = 1;
            const double ref = (double)p, got = P[(size_t)s * S + c];
            if (ref > 1e-290) {
                const double rel = fabs(got - ref) / ref;
                if (rel > worstA) worstA = rel;
                n_cmp++;
            } else if (got > 1e-289) {
                worstA = INFINITY;
            }
        }
    }

    /* (B) s-fold convolution of the one-individual distribution, quad precision, truncated at M */
    __float128 *p1 = (__float128 *)malloc(sizeof(__float128) * S);
    __float128 *cur = (__float128 *)malloc(sizeof(__float128) * S);
    __float128 *nxt = (__float128 *)malloc(sizeof(__float128) * S);
    {
        /* alpha, beta from the rates in quad precision (not from the doubles above): independent of :255-262's rounding */
        __float128 aq, bq;
        if (mu < 0 || lambda == mu) {
            aq = (__float128)lambda * t / (1 + (__float128)lambda * t);
            bq = aq;
        } else {
            const __float128 e = expq(((__float128)lambda - mu) * t);
            aq = mu * (e - 1) / (lambda * e - mu);
            bq = lambda * (e - 1) / (lambda * e - mu);
        }
        p1[0] = aq;
        __float128 pw = 1;
        for (int c = 1; c <= M; c++) {
            p1[c] = (1 - aq) * (1 - bq) * pw;
            pw *= bq;
        }
    }
    double worstB = 0;
    for (int c = 0; c <= M; c++) cur[c] = p1[c];
    for (int s = 1; s <= M; s++) {
        if (s > 1) {
            for (int c = 0; c <= M; c++) {
                __float128 acc = 0;
                for (int k = 0; k <= c; k++) acc += cur[k] * p1[c - k];
                nxt[c] = acc;
            }
            __float128 *tmp = cur;
            cur = nxt;
            nxt = tmp;
        }
        for (int c = 0; c <= M; c++) {
            const double ref = (double)cur[c], got = P[(size_t)s * S + c];
            if (ref > 1e-250) {   /* below that the Lanczos error of a 500-term lnC difference is no longer relative-small */
                const double rel = fabs(got - ref) / ref;
                if (rel > worstB) worstB = rel;
            }
        }
    }
    printf("A %.3e B %.3e rows %ld\n", worstA, worstB, n_cmp);
    free(P);
    free(lnc);
    free(p1);
    free(cur);
    free(nxt);
    return 0;
}

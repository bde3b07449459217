// host_math.hpp -- host-side scalar maths of the product (NOT the oracle):
// the ln C(n,k) tables and per-key birth-death scalars the device kernels consume.
// Each function states the reference behaviour it must reproduce (file:line).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace cafehip {

// Lanczos (g = 5, 6 terms) ln Gamma with the reference's coefficients:
// libcommon/mathfunc.c:87-89 (coefficients), :112-119 (series + closed form).
inline double gammaln(double a)
{
    static const double q[7] = {1.000000000190015,  76.18009172947146,  -86.50532032941677,
                                24.01409824083091,  -1.231739572450155, 1.208650973866179e-3,
                                -5.395239384953e-6};
    const double sqrt_2pi = 2.5066282746310002416123552393401042;
    double p = q[0];
    const double shifted = a + 5.5;
    for (int n = 1; n <= 6; ++n) p += q[n] / (a + n);
    return (a + 0.5) * std::log(shifted) - shifted + std::log(sqrt_2pi * p / a);
}

// ln C(n, r): libcommon/mathfunc.c:224-229 (r == 0 -> 0; n <= 0 or r <= 0 -> -inf).
inline double chooseln(double n, double r)
{
    if (r == 0) return 0.0;
    if (n <= 0 || r <= 0) return -INFINITY;
    return gammaln(n + 1) - gammaln(r + 1) - gammaln(n - r + 1);
}

// The two slices of the reference's chooseln cache (libtree/chooseln_cache.h:16-41)
// that the birth-death sum reads (libtree/birthdeath.c:34-73), laid out so that a
// matrix row s needs two contiguous runs:
//   A[s*ld + j] = ln C(s, j)            0 <= j <= s <= M      (first factor)
//   B[s*ld + i] = ln C(s-1+i, s-1)      1 <= s <= M, 0 <= i <= M  (second factor, i = c - j)
// ld is odd (LDS bank spreading for 16 consecutive rows).
struct LnCTables {
    int M = -1;
    int ld = 0;
    std::vector<double> A, B;
    // exp() of the two tables for the product-form matrix kernel; usable only while every
    // EA * EB * 2 stays finite (M up to ~300)
    std::vector<double> EA, EB;
    bool product_form_ok = false;
    double log2_max_prod = 0;  // log2(max EA) + log2(max EB): head-room test for the blocked matrix kernel
    void build(int M_)
    {
        M = M_;
        ld = (M + 1) | 1;
        A.assign((size_t)(M + 1) * ld, 0.0);
        B.assign((size_t)(M + 1) * ld, 0.0);
        EA.assign(A.size(), 0.0);
        EB.assign(B.size(), 0.0);
        double maxA = 0, maxB = 0;
        for (int s = 0; s <= M; ++s) {
            for (int j = 0; j <= s; ++j) {
                A[(size_t)s * ld + j] = chooseln(s, j);
                EA[(size_t)s * ld + j] = std::exp(A[(size_t)s * ld + j]);
                maxA = std::max(maxA, EA[(size_t)s * ld + j]);
            }
            if (s >= 1)
                for (int i = 0; i <= M; ++i) {
                    B[(size_t)s * ld + i] = chooseln(s - 1 + i, s - 1);
                    EB[(size_t)s * ld + i] = std::exp(B[(size_t)s * ld + i]);
                    maxB = std::max(maxB, EB[(size_t)s * ld + i]);
                }
        }
        product_form_ok = std::isfinite(maxA) && std::isfinite(maxB) && (std::log10(maxA) + std::log10(maxB) < 300.0);
        log2_max_prod = product_form_ok ? std::log2(std::max(maxA, 1.0)) + std::log2(std::max(maxB, 1.0)) : 0.0;
    }
};

// Per-key scalars of compute_birthdeath_rates (libtree/birthdeath.c:238-286):
//   mode 0: coeff <= 0  -> rows 1..M all zero        (init_zero_matrix :184-193)
//   mode 1: coeff == 1  -> identity                   (init_identity_matrix :195-208)
//   mode 2: mu < 0      -> birthdeath_rate_with_log_alpha       (:52-73)
//   mode 3: mu >= 0     -> birthdeath_rate_with_log_alpha_beta  (:34-50), also when lambda == mu >= 0 (:272-275)
struct KeyScalars {
    double log_alpha, log_beta, log_coeff, coeff;
    int mode;
    // product form: w_j = alpha^(s+c-2j) coeff^j (mode 2) or alpha^(s-j) beta^(c-j) coeff^j (mode 3)
    //             = 2^(s*l2a + c*l2b) * rho^j,  rho = coeff / (alpha * beta) = rho_m * 2^rho_e, rho_m in [1,2)
    double l2a, l2b, rho_m;
    int rho_e;
    int fast_ok;
};

inline KeyScalars key_scalars(int branchlength, double lambda, double mu)
{
    const double t = (double)branchlength;
    double alpha, beta, coeff;
    if (mu < 0 || lambda == mu) {
        alpha = lambda * t / (1 + lambda * t);
        beta = alpha;
        coeff = 1 - 2 * alpha;
    } else {
        const double e_diff = std::exp((lambda - mu) * t);
        const double numerator = e_diff - 1;
        const double denominator = lambda * e_diff - mu;
        alpha = (mu * numerator) / denominator;
        beta = (lambda * numerator) / denominator;
        coeff = 1 - alpha - beta;
    }
    KeyScalars k;
    k.coeff = coeff;
    if (!(coeff > 0)) {
        // coeff <= 0 (a NaN coeff fails every reference test `coeff <= 0`, `coeff == 1`,
        // `coeff > 0 && coeff != 1`, leaving the calloc'd zero rows: same as mode 0)
        k.mode = 0;
        k.log_alpha = k.log_beta = k.log_coeff = 0;
        k.l2a = k.l2b = 0; k.rho_m = 1; k.rho_e = 0; k.fast_ok = 0;
        return k;
    }
    if (coeff == 1) {
        k.mode = 1;
        k.log_alpha = k.log_beta = k.log_coeff = 0;
        k.l2a = k.l2b = 0; k.rho_m = 1; k.rho_e = 0; k.fast_ok = 0;
        return k;
    }
    k.log_alpha = std::log(alpha);
    k.log_beta = std::log(beta);
    k.log_coeff = std::log(coeff);
    k.mode = (mu < 0) ? 2 : 3;
    k.l2a = std::log2(alpha);
    k.l2b = std::log2(beta);
    const double rho = coeff / (alpha * beta);
    k.fast_ok = std::isfinite(rho) && rho > 0 && std::isfinite(k.l2a) && std::isfinite(k.l2b);
    if (k.fast_ok) {
        k.rho_e = std::ilogb(rho);
        k.rho_m = std::scalbn(rho, -k.rho_e);
    } else {
        k.rho_e = 0;
        k.rho_m = 1.0;
    }
    return k;
}

}  // namespace cafehip

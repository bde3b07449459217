// exp_like_host.hpp -- exp() as THIS HOST's libm computes it, on the device.
//
// The report phase compares matrix entries with == and < (cafe/viterbi.cpp:60-67) and uniform draws with cumulative row sums
// (cafe/cafe_tree.c:533-569), so its matrices should carry the bits the reference's build carries on the same machine.  K1's
// exact form already runs the reference's operation sequence (libtree/birthdeath.c:34-73) with contraction off; the one
// operation that was not the host's is exp(): with the device library's, 94.2 % of the entries were bit-identical and the rest
// up to 3 ulp off (round 4 measurement).  glibc >= 2.28 computes exp() with the table-driven algorithm of ARM's optimized
// routines (N = 128: x = k ln2/128 + r, 2^(k/128) from a table as scale (1 + tail), exp(r) - 1 by a degree-5 polynomial), a
// fixed sequence of IEEE operations -- restated here; x86-64 glibc selects at load time a build of it compiled with or without
// fused multiply-add, hence the two forms.  Which one this host runs (or neither: any other libm) is DETECTED when the library
// is first used, by comparing both forms with std::exp on 200,000 arguments; K1's exact form then calls the matching one, or the
// device library's exp when neither matches.  The table is computed in quad precision by tools/exp_table_gen.c;
// tests/test_exp_like_host.py checks the host build of this function against the host's exp() on 10^7 arguments, bit for bit.
//
// Source of the algorithm (third party, NOT the CAFE reference): `exp` of ARM's Optimized Routines (math/exp.c, Copyright (c)
// 2018 Arm Limited, SPDX-License-Identifier: MIT -- later releases: MIT OR Apache-2.0 WITH LLVM-exception), adopted by glibc
// 2.28 as sysdeps/ieee754/dbl-64/e_exp.c (LGPL-2.1-or-later).  The reduction constants, the polynomial coefficients, the
// table layout (tail, scale bits) and the special-case ladder below are that algorithm's -- they have to be, bit for bit,
// or the results would not be the host's; the code is restated for host + device and the table regenerated in quad
// precision (tools/exp_table_gen.c), not copied from either tree.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

namespace cafehip {

static const uint64_t kExpTabHost[256] = {
#include "exp_like_host_table.inc"
};
static __device__ const uint64_t kExpTabDev[256] = {
#include "exp_like_host_table.inc"
};

__host__ __device__ inline uint64_t exp_bits(double d)
{
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
__host__ __device__ inline double exp_from_bits(uint64_t u)
{
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// FUSED: the build of the algorithm compiled with fused multiply-add (what x86-64 glibc runs on a CPU that has it)
template <bool FUSED>
__host__ __device__ inline double exp_like_host(double x)
{
    // every fused operation below is written as fma(): the compiler must not form others (scoped to this body: the
    // including translation units keep their own contraction setting)
#pragma clang fp contract(off)
#ifdef __HIP_DEVICE_COMPILE__
    const uint64_t* const T = kExpTabDev;
#else
    const uint64_t* const T = kExpTabHost;
#endif
    const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double Shift = 0x1.8p52;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const uint32_t abstop = (uint32_t)(exp_bits(x) >> 52) & 0x7ff;
    bool special = false;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {            // |x| < 2^-54 or |x| >= 512
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;   // tiny: the result is 1 up to rounding
        if (abstop >= 0x409u) {                           // |x| >= 1024, inf, nan
            if (exp_bits(x) == exp_bits(-INFINITY)) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (exp_bits(x) >> 63) ? 0.0 : INFINITY;
        }
        special = true;                                   // 512 <= |x| < 1024: the scale may leave the double range
    }
    const double z = InvLn2N * x;
    double kd = z + Shift;
    const uint64_t ki = exp_bits(kd);
    kd -= Shift;
    const double r = FUSED ? fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x)) : x + kd * NegLn2hiN + kd * NegLn2loN;
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = ki << 45;
    const double tail = exp_from_bits(T[idx]);
    uint64_t sbits = T[idx + 1] + top;
    const double r2 = r * r;
    double tmp;
    if (FUSED) {
        const double p1 = fma(r, C3, C2), p2 = fma(r, C5, C4);
        tmp = fma(r2 * r2, p2, fma(r2, p1, tail + r));
    } else {
        tmp = tail + r + r2 * (C2 + r * C3) + r2 * r2 * (C4 + r * C5);
    }
    if (special) {
        if ((ki & 0x80000000u) == 0) {   // k > 0
            sbits -= 1009ull << 52;
            const double scale = exp_from_bits(sbits);
            return 0x1p1009 * (FUSED ? fma(scale, tmp, scale) : scale + scale * tmp);
        }
        sbits += 1022ull << 52;          // k < 0: care in the subnormal range.  The product is used twice and is kept unfused in
                                         // BOTH forms here; a libm build that fused one use anyway would differ in this
                                         // range, which the detection at first use samples (x < -708): it would be rejected
        const double scale = exp_from_bits(sbits);
        const double prod = scale * tmp;
        double y = scale + prod;
        if (y < 1.0) {
            double lo = scale - y + prod;
            const double hi = 1.0 + y;
            lo = 1.0 - hi + y + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    const double scale = exp_from_bits(sbits);
    return FUSED ? fma(scale, tmp, scale) : scale + scale * tmp;
}

// 1: this host's exp() is the fused form, 2: the plain form, 0: neither (the device library's exp is used).  Decided once.
inline int host_exp_variant(long* mismatches_fused = nullptr, long* mismatches_plain = nullptr, long n = 200000, unsigned seed = 12345)
{
    uint64_t s = 0x9E3779B97F4A7C15ull ^ seed;
    long bad1 = 0, bad2 = 0;
    for (long i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(s >> 11) * 0x1p-53;
        double x;
        switch (i & 3) {
            case 0: x = -60.0 * u; break;                 // where the terms of a transition matrix live
            case 1: x = -745.2 + 745.2 * u; break;
            case 2: x = -745.2 + 240.0 * u; break;        // the special range below -512
            default: x = -1.0 + 2.0 * u; break;
        }
        const volatile double xv = x;                     // (the library call, not a compile-time constant)
        const double want = std::exp(xv);
        bad1 += exp_bits(exp_like_host<true>(x)) != exp_bits(want);
        bad2 += exp_bits(exp_like_host<false>(x)) != exp_bits(want);
    }
    if (mismatches_fused) *mismatches_fused = bad1;
    if (mismatches_plain) *mismatches_plain = bad2;
    return bad1 == 0 ? 1 : (bad2 == 0 ? 2 : 0);
}

}  // namespace cafehip

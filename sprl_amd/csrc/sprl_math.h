// sprl_math.h — deterministic float log / exp / pow shared by host C and gfx950 device code.
//
// Why this exists: the reference calls libm's logf / powf / expf from four places on the
// self-play path (libstdc++ gamma_distribution inside Random::Dirichlet, utils/random.cpp:61-74;
// GameActionDist::pow for the tempered visit pdf, games/GameActionDist.hpp:113-119 via
// selfplay/SelfPlay.hpp:115-119; GameActionDist::exp in networks/GridNetwork.hpp:110).
// glibc's float routines are not available on the device and are not correctly rounded, so a
// GPU engine cannot reproduce their last bit.  These routines compute in IEEE double with
// +,-,*,/ only (no FMA contraction, no tables, no libm) and round once to float: the same
// source gives the same bits with gcc on the host and hipcc on gfx950, and the float result is
// the correctly rounded one except when the exact value lies within ~1e-16 relative of a
// rounding boundary.  tests/test_math.py bounds the distance to glibc at 1 float ulp.
//
// The double kernels follow the classical argument-reduction + minimax-polynomial scheme
// (k*ln2 split, s = f/(2+f) series for log; r - k*ln2 and the c = r - r^2*P(r^2) form for exp).
#ifndef SPRL_MATH_H
#define SPRL_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SPRL_HD __host__ __device__ static inline
#else
#define SPRL_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif

SPRL_HD uint64_t sprl_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
SPRL_HD double sprl_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
SPRL_HD uint32_t sprl_f2u(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
SPRL_HD float sprl_u2f(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }

// Natural log of a positive, finite, normal double (every float > 0 converts to one).
// x <= 0 returns -inf (x == 0) or NaN (x < 0); inf/NaN pass through.
SPRL_HD double sprl_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = sprl_d2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    if ((u << 1) == 0) return -1.0 / 0.0;
    if (hx >> 31) return (x - x) / 0.0;
    if (hx >= 0x7ff00000u) return x;
    int k = 0;
    if (hx < 0x00100000u) {            // subnormal double: scale up (unreachable from float inputs)
        k -= 54;
        x *= 18014398509481984.0;
        u = sprl_d2u(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    // reduce x into [sqrt(2)/2, sqrt(2))
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    x = sprl_u2d(u);

    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// e^x for finite x; saturates to +inf above 709.78 and to 0 below -745.
SPRL_HD double sprl_exp(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return 1.0 / 0.0;
    if (x < -745.13321910194110842) return 0.0;
    double ax = x < 0 ? -x : x;
    double hi, lo;
    int k;
    if (ax > 0.34657359027997264) {    // |x| > 0.5 ln2
        if (ax >= 1.0397207708399179)  // |x| >= 1.5 ln2
            k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        else
            k = x < 0 ? -1 : 1;
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax > 3.725290298461914e-09) {  // 2^-28
        k = 0;
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    double xx = x * x;
    double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    // y * 2^k with k in [-1075, 1024]; split so that each factor is a normal double.
    int k1 = k / 2, k2 = k - k1;
    double s1 = sprl_u2d((uint64_t)(0x3ff + k1) << 52);
    double s2 = sprl_u2d((uint64_t)(0x3ff + k2) << 52);
    return y * s1 * s2;
}

SPRL_HD float sprl_logf(float x) { return (float)sprl_log((double)x); }

SPRL_HD float sprl_expf(float x) { return (float)sprl_exp((double)x); }

// x^y for x >= 0 (the only domain the self-play path uses: pdf entries and uniform draws).
SPRL_HD float sprl_powf(float x, float y) {
    if (y == 0.0f || x == 1.0f) return 1.0f;
    if (x == 0.0f) return y > 0.0f ? 0.0f : 1.0f / 0.0f;
    if (x < 0.0f || x != x || y != y) return (x - x) / (x - x);
    return (float)sprl_exp((double)y * sprl_log((double)x));
}

#if !defined(__clang__) && defined(__GNUC__)
#pragma GCC pop_options
#endif

#endif  // SPRL_MATH_H

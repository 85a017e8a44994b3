// vigo_exact_pow.hpp — the correctly rounded integer power t^d, the platform-independent reading of the
// reference's `pow(t, d)` in polyTrajSolver::getPose (polyTrajSolver.cpp:1035-1039).
//
// The reference calls libm's pow(double, double) with an integer-valued exponent.  glibc >= 2.28 returns the
// correctly rounded power in 99.9 % of the cases and its neighbour otherwise (measured here: 9e-4 of random
// (t, d) pairs differ from the exact result, for d = 2 as well; tools/pow_rounding_rate.py), and which
// neighbour depends on the libm build (FMA or not), so "the reference's bits" are not a function of the inputs
// alone.  The device therefore evaluates THE correctly rounded power, which every conforming libm approximates
// to within one unit in the last place, in two tiers:
//
//   pow_step()    a running double-double product (hi, lo) <- (hi, lo) * t: fma-exact product, one rounding of
//                 the low part, fast two-sum.  Relative error <= 2^-105 per step, < 2^-101 after the 14 steps of
//                 a degree-15 polynomial.  RN(hi + lo) = hi is the correctly rounded power unless the true
//                 value lies within that error of a rounding boundary; the step reports `ambiguous` whenever
//                 |lo| is within 2^-40 (relative) of half an ulp of hi — a superset of those cases, probability
//                 2^-40 per power — and for results outside [2^-900, 2^900], where the low parts would underflow;
//   pow_exact()   exact integer arithmetic on the significand (<= 795 bits), rounded once to nearest-even,
//                 subnormal results and overflow included.  Taken for the ambiguous cases only.
//
// Compiled for the host too: vigo_exact_pow() of the C ABI exposes pow_exact(), and tests/test_abi.py pins both
// tiers against Python's exact rational arithmetic.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VIGO_PHD __host__ __device__ __forceinline__
#define VIGO_PHD_NOINLINE inline __host__ __device__ __noinline__
#else
#define VIGO_PHD inline
#define VIGO_PHD_NOINLINE inline
#endif

namespace vigo {

VIGO_PHD uint64_t pow_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

VIGO_PHD int pow_clz64(uint64_t v) {   // v != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}

// RN-even of t^d for any double t and 0 <= d <= 15, by exact integer arithmetic.
VIGO_PHD_NOINLINE double pow_exact(double t, int d) {
    if (d <= 0) return 1.0;                       // pow(x, 0) = 1 for every x, NaN included (C99 F.9.4.4)
    if (d == 1 || !(fabs(t) <= 1.79769313486231570815e308) || t == 0.0) {
        // exact cases and the IEEE specials (NaN, +-inf, +-0): repeated multiplication gives pow()'s answer
        double r = t;
        for (int i = 1; i < d; ++i) r *= t;
        return r;
    }
    uint64_t bits;
    {
        const double a = fabs(t);
        memcpy(&bits, &a, sizeof(bits));
    }
    int e = (int)(bits >> 52);
    uint64_t m = bits & (((uint64_t)1 << 52) - 1);
    if (e == 0) e = 1; else m |= (uint64_t)1 << 52;      // |t| = m * 2^(e - 1075)
    int ex = e - 1075;
    while ((m & 1u) == 0) { m >>= 1; ++ex; }              // m odd: keeps the product short
    // big = m^d, little-endian 64-bit limbs (53 * 15 = 795 bits at most)
    uint64_t L[13];
    for (int i = 0; i < 13; ++i) L[i] = 0;
    L[0] = 1;
    int n = 1;
    for (int i = 0; i < d; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < n; ++j) {
            const uint64_t lo = L[j] * m, hi = pow_mulhi64(L[j], m);
            const uint64_t s = lo + carry;
            carry = hi + (s < lo ? 1u : 0u);
            L[j] = s;
        }
        if (carry && n < 13) L[n++] = carry;
    }
    const long long E = (long long)ex * d;                // t^d = big * 2^E
    const int top = 64 * (n - 1) + 63 - pow_clz64(L[n - 1]);   // index of the leading bit
    const long long exp2 = top + E;                       // t^d in [2^exp2, 2^(exp2+1))
    const bool neg = (t < 0.0) && (d & 1);
    if (exp2 > 1023) return neg ? -INFINITY : INFINITY;
    // significand bits that survive: 53 for a normal result, fewer for a subnormal one
    long long keep = 53;
    if (exp2 < -1022) keep = 53 - (-1022 - exp2);
    if (keep < 0) return neg ? -0.0 : 0.0;                // below half the smallest subnormal
    const long long shift = (long long)top + 1 - keep;    // low bits dropped (may be <= 0: nothing dropped)
    uint64_t q = 0;
    bool round = false, sticky = false;
    if (shift <= 0) {
        q = L[0];                                          // top < 53: the whole integer, exact
    } else {
        auto bit = [&](long long i) -> unsigned { return (unsigned)((L[i >> 6] >> (i & 63)) & 1u); };
        for (long long i = top; i >= shift; --i) q = (q << 1) | bit(i);
        round = bit(shift - 1) != 0;
        for (long long i = shift - 2; i >= 0 && !sticky; --i) sticky = bit(i) != 0;
        if (round && (sticky || (q & 1u))) ++q;
    }
    // q <= 2^53 and q * 2^(E + max(shift, 0)) is representable by construction: ldexp is exact
    const double r = ldexp((double)q, (int)(E + (shift > 0 ? shift : 0)));
    return neg ? -r : r;
}

// One step of the running double-double power: (hi, lo) <- (hi, lo) * t.  Returns true when the rounding of
// hi + lo to hi cannot be certified (see the header comment); hi is then NOT guaranteed to be the correctly
// rounded power and the caller re-evaluates with pow_exact().
VIGO_PHD bool pow_step(double& hi, double& lo, double t) {
    const double p = hi * t;
    const double e = fma(lo, t, fma(hi, t, -p));
    const double s = p + e;
    const double l = e - (s - p);
    hi = s;
    lo = l;
    const double a = fabs(s);
    const bool range_ok = a >= 0x1p-900 && a <= 0x1p900;           // false for NaN and +-inf too
    return !(range_ok && (s + l * (1.0 + 0x1p-40) == s)) && t != 0.0;
}

}  // namespace vigo

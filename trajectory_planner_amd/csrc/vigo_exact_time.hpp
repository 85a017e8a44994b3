// vigo_exact_time.hpp — closed-form evaluation of the reference's accumulated sample clock.
//
// The reference samples trajectories with `for (double t = 0; t < end; t += delT)`
// (polyTrajSolver.cpp:1129, polyTrajOctomap.cpp:638-653), i.e. t_k is the k-fold floating
// point sum fl(fl(0 + d) + d) ..., NOT k*d.  To sample in parallel and still see the same
// bits, accumulated_time(d, k) returns exactly that t_k in O(#binades) steps instead of k:
//
//   while t and fl(t + d) stay in one binade [2^e, 2^(e+1)), t is a multiple of
//   u = 2^(e-52) and fl(t + d) = t + R(d), R = d rounded to a multiple of u (round-to-nearest,
//   a tie resolved towards an even result mantissa).  R is constant except that in the tie
//   case it may differ on the first step after entering the binade (until the mantissa is
//   even).  So: take real steps until two have landed in the current binade, measure
//   inc = fl(t + d) - t (exact), jump j = floor((2^53 - 1 - t/u) / (inc/u)) steps in exact
//   integer arithmetic, and let real steps carry t across the binade boundary.
//
// Compiled for host too (tests/ pin it against the sequential loop).
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VIGO_HD __host__ __device__ __forceinline__
#else
#define VIGO_HD inline
#endif

namespace vigo {

VIGO_HD double accumulated_time(double d, int64_t k) {
    double t = 0.0;
    if (!(d > 0.0) || !(d < 1e300)) {  // degenerate clocks: plain recurrence
        for (int64_t i = 0; i < k; ++i) t = t + d;
        return t;
    }
    int64_t i = 0;
    int e_prev = -100000;
    int landed = 0;
    while (i < k) {
        t = t + d;
        ++i;
        const int e = ilogb(t);
        if (e != e_prev) { e_prev = e; landed = 1; } else { ++landed; }
        if (landed >= 2 && i < k) {
            const double tn = t + d;
            if (ilogb(tn) == e) {
                const double inc = tn - t;  // exact: both multiples of 2^(e-52), same binade
                if (inc == 0.0) return t;   // d below half an ulp: the clock has stalled for good
                const int64_t tm = (int64_t)ldexp(t, 52 - e);      // in [2^52, 2^53)
                const int64_t im = (int64_t)ldexp(inc, 52 - e);    // >= 1
                if (im >= 1) {
                    int64_t j = (((int64_t)1 << 53) - 1 - tm) / im;  // t + j*inc <= 2^(e+1) - u
                    if (j > k - i) j = k - i;
                    if (j > 0) {
                        t = ldexp((double)(tm + j * im), e - 52);
                        i += j;
                    }
                }
            }
        }
    }
    return t;
}

// ---- the sample clock as a table (k_corridor) ---------------------------------------------------------------
// accumulated_time(d, k) costs O(#binades) steps with a 64-bit division each — about as much as four samples.  All
// samples of a segment share d, so ONE thread writes the closed form down while it computes the last clock value: the
// clock is piecewise affine in k, exactly — inside a binade t_k = t_k0 + (k - k0) * inc with t_k0, inc multiples of the
// binade's ulp and the sum below 2^53 ulps, so the product and the sum are exact in fp64 — with one piece per run of
// equal increments and one per real step in between (<= 4 per binade).  clock_at() is then a binary search.
constexpr int kClockCap = 128;
struct ClockTable {
    int n;                       // pieces, or -1: no table (degenerate clock, more pieces than kClockCap)
    int k0[kClockCap];
    double t0[kClockCap], inc[kClockCap];
};

// binary exponent of a normal positive double (== ilogb there)
VIGO_HD int clock_exponent(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return (int)((u >> 52) & 0x7ff) - 1023;
}

// Thread-serial; returns t_{k_last} == accumulated_time(d, k_last), whose steps these are with cheaper arithmetic (the
// table is built once per segment by ONE lane while 255 wait): the binade from the exponent field, and the jump
// j = floor((top - t) / inc), top = the last double of the binade, by one fp64 division corrected with the exact
// residual fma(-j, inc, top - t) (top - t, inc and the residual are multiples of the binade's ulp below 2^53 of them)
// instead of a 64-bit integer division.  Normal clocks only: d in [2^-1000, 1e300), t below 1e300; otherwise no table.
VIGO_HD double build_clock_table(double d, int k_last, ClockTable& C) {
    C.n = -1;
    if (!(d >= 0x1p-1000) || !(d < 1e300)) return accumulated_time(d, k_last);
    int m = 0;
    auto put = [&](int k, double tk) {
        if (m < kClockCap) { C.k0[m] = k; C.t0[m] = tk; C.inc[m] = 0.0; }
        ++m;
    };
    double t = 0.0;
    int i = 0, e_prev = -100000, landed = 0;
    put(0, 0.0);
    while (i < k_last) {
        t = t + d;
        ++i;
        put(i, t);
        if (!(t < 1e300)) return accumulated_time(d, k_last);
        const int e = clock_exponent(t);
        if (e != e_prev) { e_prev = e; landed = 1; } else { ++landed; }
        if (landed >= 2 && i < k_last) {
            const double tn = t + d;
            if (clock_exponent(tn) == e) {
                const double inc = tn - t;  // exact: both multiples of 2^(e-52), same binade
                if (inc == 0.0) break;      // the clock has stalled for good: the last piece (inc 0) covers the rest
                const uint64_t top_bits = ((uint64_t)(e + 1023) << 52) | 0xfffffffffffffull;
                double top;
                memcpy(&top, &top_bits, sizeof top);
                const double num = top - t;                     // exact
                double q = floor(num / inc);
                const double r = fma(-q, inc, num);             // exact
                if (r < 0.0) q -= 1.0;
                else if (r >= inc) q += 1.0;
                const int left = k_last - i;
                const int j = q >= (double)left ? left : (int)q;
                if (j > 0) {
                    if (m <= kClockCap) C.inc[m - 1] = inc;     // the piece of step i runs on for j more steps
                    t = t + (double)j * inc;                    // exact (j * inc and the sum stay below 2^53 ulps)
                    i += j;
                }
            }
        }
    }
    if (m <= kClockCap) C.n = m;
    return t;
}

// the last piece that starts at or before k
VIGO_HD int clock_piece(const ClockTable& C, int k) {
    int lo = 0, hi = C.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (C.k0[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// t_k given a piece at or before k's own (pieces are thousands of samples long: the loop rarely turns)
VIGO_HD double clock_from(const ClockTable& C, int piece, int k) {
    while (piece + 1 < C.n && C.k0[piece + 1] <= k) ++piece;
    return C.t0[piece] + (double)(k - C.k0[piece]) * C.inc[piece];
}
VIGO_HD double clock_at(const ClockTable& C, int k) { return clock_from(C, clock_piece(C, k), k); }

}  // namespace vigo

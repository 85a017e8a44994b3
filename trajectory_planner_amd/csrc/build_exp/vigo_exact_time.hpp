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

}  // namespace vigo

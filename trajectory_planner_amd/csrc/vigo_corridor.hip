// vigo_corridor.hip — min-snap corridor collision checker (polyTrajOctomap::checkCollisionTraj
// -> checkCollision -> checkCollisionPoint, PO.cpp:547-589, :634-656) fed by the polynomial
// sampler (polyTrajSolver::getPose, PS.cpp:1026-1056), plus the trilinear ESDF query.
//
// One 256-thread workgroup per polynomial segment:
//   bound   the segment's Bernstein coefficients over its sampled time span bound every sample
//           position (convex hull) — no pass over the samples;
//   stage   the voxels that box (+ the collision box + 1 voxel) can touch are copied from the
//           packed HBM planes into an LDS tile as ONE bit per voxel (unknown | occupied — both
//           mean "collides" for the sweep, PO.cpp:580-588);
//   decide  spans of 64 / 32 / 16 consecutive samples by ONE evaluation each where that provably gives every sample's
//           verdict (an interval that holds all their float positions, pushed through the reference's own monotone
//           expressions at both ends: same voxel keys at both ends = same keys for every sample), spans that cannot be
//           decided cut in four, twice; what is left goes through the per-sample sweep, compacted so that every lane
//           has a sample (SpanConst, k_corridor PASS 0).  The sample clock t += delT comes from a per-segment table
//           (vigo_exact_time.hpp);
//   walk    segments the certificates do not apply to (degenerate clocks, boxes of more than 3 map cells per axis,
//           non-finite coefficients, samples further apart than 1/32 of a voxel) take the walk of rounds 1-2: every
//           thread over chunks of 16 consecutive samples, every lattice point of the box looked up in the tile
//           (k_corridor PASS 1).
// A tile too large for the LDS budget falls back to lookups in the packed planes (L2).
#include <type_traits>

#include "vigo_corridor.hpp"
#include "vigo_exact_pow.hpp"
#include "vigo_exact_time.hpp"
#include "vigo_grid.hpp"

namespace vigo {
namespace {

constexpr int kChunk = 16;       // consecutive samples per thread visit
constexpr int kBlock = 256;
constexpr int kMaxDeg = 15;
constexpr int kQueueCap = 512;    // LDS queue of samples for the exact-power pass of k_corridor

// collision_box / map_resolution of the sweep and octomap's resolution_factor (1 / tree resolution)
struct SweepConst {
    double box[3], map_res, rf;
};

struct CorridorArgs {
    int S, deg;
    const double* coeffs;
    const int32_t* n_samp;
    const double* delT;
    uint8_t* out_flag;
    int32_t* out_first;
    int32_t* out_count;
    int tile_words_cap;
    int* todo;           // per segment: 1 = left to the second pass (see k_corridor)
    const ClockTable* clocks;   // per segment, written by k_corridor_clocks; NULL: every workgroup builds its own
    SweepConst sweep;
};

// ---- the sampler: polyTrajSolver::getPose, PS.cpp:1035-1039 -------------------------------------------
//   x += c[d] * pow(t, d), d ascending, for the three axes.
// pow(t, d) is evaluated as THE correctly rounded power (vigo_exact_pow.hpp: running double-double product,
// certified per power, exact integer arithmetic for the 2^-40 of cases that cannot be certified), so the sample
// positions — and with them the float coordinates, voxel keys, flags and indices the checker derives — are a
// function of the inputs alone, whatever libm the reference was linked against.
__device__ __forceinline__ void poly_pos(const double* cf, int deg, double t, double (&p)[3]) {
    double x = 0, y = 0, z = 0, hi = 1.0, lo = 0.0;
    bool amb = false;
    for (int d = 0; d <= deg; ++d) {
        if (d == 1) hi = t;
        else if (d > 1) amb |= pow_step(hi, lo, t);
        x += cf[d] * hi;
        y += cf[(kMaxDeg + 1) + d] * hi;
        z += cf[2 * (kMaxDeg + 1) + d] * hi;
    }
    if (__any(amb)) {
        if (amb) {
            x = y = z = 0;
            for (int d = 0; d <= deg; ++d) {
                const double pw = pow_exact(t, d);
                x += cf[d] * pw;
                y += cf[(kMaxDeg + 1) + d] * pw;
                z += cf[2 * (kMaxDeg + 1) + d] * pw;
            }
        }
    }
    p[0] = x; p[1] = y; p[2] = z;
}

// the planner's degree (cfg polynomial_degree: 7): coefficients in registers, same operation order
__device__ __forceinline__ void poly_pos7(const double (&c7)[3][8], double t, double (&p)[3]) {
    double pw[8];
    pw[0] = 1.0;
    pw[1] = t;
    double hi = t, lo = 0.0;
    bool amb = false;
#pragma unroll
    for (int d = 2; d < 8; ++d) {
        amb |= pow_step(hi, lo, t);
        pw[d] = hi;
    }
    if (__any(amb)) {
        if (amb) {
#pragma unroll
            for (int d = 2; d < 8; ++d) pw[d] = pow_exact(t, d);
        }
    }
    double x = 0, y = 0, z = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        x += c7[0][d] * pw[d];
        y += c7[1][d] * pw[d];
        z += c7[2][d] * pw[d];
    }
    p[0] = x; p[1] = y; p[2] = z;
}

// ---- the float position the checker consumes (pose2Octomap, PO.cpp:634-656), by a floating-point filter ----
// The box sweep sees (float)x, (float)y, (float)z only.  Evaluating the exact-power chain above for every sample
// costs 30 % of the whole checker (0.685 -> 0.895 ms on config 3), so the kernels use the classical
// filtered-predicate scheme instead:
//   fast   the same sums with the powers formed by repeated multiplication (1 instead of ~10 operations per power);
//   bound  |fast - exact chain| <= (3 deg + 2) u A with u = 2^-53 and A = sum_d |c_d| T^d >= every partial sum and
//          term (T = the largest sampled |t|): d u |c_d t^d| from the power, 2 u |c_d t^d| from the two roundings of
//          the product, 2 u A per addition.  E = 2^-46 A + 2^-1000 (= 128 u A, more than twice the bound for
//          deg <= 15, plus the absolute error of a gradual underflow) is computed once per segment and axis;
//   filter the conversion to float is monotone, so (float)(fast - E) == (float)(fast + E) certifies that value as the
//          float of the exact chain.  A sample that cannot certify all three axes (probability ~2^-21 A / |x| per
//          axis, or a NaN / overflow anywhere) is re-evaluated with the exact chain: k_corridor queues its index in
//          LDS and sweeps the queue after the main pass, so the rare path costs the main loop no registers.
// Results are therefore those of the exact-power chain by construction.
__device__ __forceinline__ double sampler_error_bound(const double* c, int deg, double t_last) {
    const double T = fabs(t_last);
    double A = 0.0, pw = 1.0;
    for (int d = 0; d <= deg; ++d) {
        A += fabs(c[d]) * pw;
        pw *= T;
    }
    return 0x1p-46 * A + 0x1p-1000;   // inf / NaN for such coefficients or clocks: nothing certifies, every sample is exact
}

__device__ __forceinline__ void poly_fast7(const double (&c7)[3][8], double t, double (&p)[3]) {
    double x = 0, y = 0, z = 0, pw = 1.0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        x += c7[0][d] * pw;
        y += c7[1][d] * pw;
        z += c7[2][d] * pw;
        pw *= t;
    }
    p[0] = x; p[1] = y; p[2] = z;
}
// (coefficients from LDS, all lanes one address: the first pass of k_corridor has no registers to hold 24 doubles)
__device__ __forceinline__ void poly_fast7_lds(const double* cf, double t, double (&p)[3]) {
    double x = 0, y = 0, z = 0, pw = 1.0;
    int off = 0;
    asm volatile("" : "+v"(off));       // (an offset the compiler cannot see through: the loads stay here, not in registers above the loop)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        x += cf[off + d] * pw;
        y += cf[off + (kMaxDeg + 1) + d] * pw;
        z += cf[off + 2 * (kMaxDeg + 1) + d] * pw;
        pw *= t;
    }
    p[0] = x; p[1] = y; p[2] = z;
}
__device__ __forceinline__ void poly_fast(const double* cf, int deg, double t, double (&p)[3]) {
    double x = 0, y = 0, z = 0, pw = 1.0;
    for (int d = 0; d <= deg; ++d) {
        x += cf[d] * pw;
        y += cf[(kMaxDeg + 1) + d] * pw;
        z += cf[2 * (kMaxDeg + 1) + d] * pw;
        pw *= t;
    }
    p[0] = x; p[1] = y; p[2] = z;
}

__device__ __forceinline__ double uniform_f64(double v) {   // a wave-uniform value through SGPRs
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// fast form + filter: true when f holds the certified floats of the exact chain
template <bool DEG7, bool REG = true>
__device__ __forceinline__ bool sample_f32_fast(const double (&c7)[3][8], const double* cf, int deg, double t,
                                                const double (&E)[3], float (&f)[3]) {
    double p[3];
    if (DEG7 && REG) poly_fast7(c7, t, p);
    else if (DEG7) poly_fast7_lds(cf, t, p);
    else poly_fast(cf, deg, t, p);
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = (float)(p[a] - E[a]), hi = (float)(p[a] + E[a]);
        f[a] = lo;
        ok = ok && (lo == hi);            // false for NaN
    }
    return ok;
}
// the exact-power chain itself (coefficients from LDS: this path is rare and must not cost the fast one registers)
__device__ __noinline__ void sample_f32_exact(const double* cf, int deg, double t, float (&f)[3]) {
    double p[3];
    poly_pos(cf, deg, t, p);
#pragma unroll
    for (int a = 0; a < 3; ++a) f[a] = (float)p[a];
}

// order-preserving float <-> int for LDS atomic min/max
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// ---- the box sweep of one pose (polyTrajOctomap::checkCollision, PO.cpp:547-568) -----------
// The reference walks the lattice xi, yi, zi and returns at the first point that is outside the
// metric bounds, outside the tree (unknown) or occupied; only the boolean leaves the function, so
// the result is the OR over all lattice points and the walk order is free.  Per axis the lattice
// coordinate, its bounds test and its voxel key depend on that axis' index alone: they are
// evaluated once per axis point (3 + 3 + 2 for the cfg box) instead of once per lattice point
// (3 + 9 + 18), with exactly the reference's expressions (float coordinates, PO.cpp:558-560;
// floor(coord * resolution_factor) keys, octomap coordToKey).
constexpr int kAxisMax = 4;  // lattice points per axis on the fast path (box / map_res <= 3)

__device__ __forceinline__ void axis_keys(double lo, int num, double map_res, double rf, double bmin, double bmax,
                                          int key0, int dim, int (&k)[kAxisMax], bool& out) {
#pragma unroll
    for (int i = 0; i < kAxisMax; ++i) {
        k[i] = 0;
        if (i <= num) {
            const float q = (float)(lo + i * map_res);
            out |= !(q >= bmin && q <= bmax);                      // PO.cpp:572-577 metric bounds (a NaN pose is outside)
            const int kk = (int)floor(rf * (double)q) - key0;
            out |= (kk < 0) || (kk >= dim);                        // no node there: unknown -> occupied
            k[i] = kk;
        }
    }
}

struct Tile {
    int x0, y0, w0;      // first voxel x, y and first z-word covered
    int tx, ty, tw;      // extent in x, y voxels and z words
    bool in_lds;
};

// Per-lane memory of the previous pose's lattice keys and verdict (k_corridor).  The verdict of the fast path is a
// function of the per-axis keys alone; consecutive samples of a segment are ~0.1 mm apart against 100 mm voxels,
// so the keys repeat for hundreds of samples and the 18 lookups are skipped whenever every lane of the wave
// repeats its keys (a wave-uniform branch: results cannot depend on it).
struct SweepMemo {
    int kx[kAxisMax], ky[kAxisMax], kz[kAxisMax];
    int nums;            // xNum | yNum << 8 | zNum << 16, -1 = nothing remembered
    bool verdict;
};

// Range of the lattice counts of a segment's poses and the dividing line between them (k_corridor, see SpanConst):
// xNum = (int)(((fx + h) - (fx - h)) / map_res) is nlo or nhi = nlo + 1 for every pose of the segment, and it is nhi
// exactly when the computed difference reaches thr (the smallest double whose quotient truncates to nhi, found per
// segment by stepping ulps around nhi * map_res) — the same integers as the division, for a compare.
struct CountConst {
    int nlo[3], nhi[3];
    double thr[3];
};

// OR over the box lattice given per-axis keys (all inside the grid): LDS tile when it holds them, packed planes otherwise
__device__ __forceinline__ bool lattice_any(const GridView& g, const Tile* T, const uint32_t* tile_words, const int (&kx)[kAxisMax],
                                            const int (&ky)[kAxisMax], const int (&kz)[kAxisMax], int xNum, int yNum, int zNum) {
    // keys grow with the lattice index: the tile holds all of them iff it holds the first and last
    const bool tiled = T && T->in_lds && kx[0] >= T->x0 && kx[xNum] < T->x0 + T->tx && ky[0] >= T->y0 &&
                       ky[yNum] < T->y0 + T->ty && (kz[0] >> 5) >= T->w0 && (kz[zNum] >> 5) < T->w0 + T->tw;
    unsigned any = 0u;
    if (tiled) {
#pragma unroll
        for (int xi = 0; xi < kAxisMax; ++xi) {
            if (xi > xNum) continue;
#pragma unroll
            for (int yi = 0; yi < kAxisMax; ++yi) {
                if (yi > yNum) continue;
                const int col = ((kx[xi] - T->x0) * T->ty + (ky[yi] - T->y0)) * T->tw - T->w0;
#pragma unroll
                for (int zi = 0; zi < kAxisMax; ++zi)
                    if (zi <= zNum) any |= tile_words[col + (kz[zi] >> 5)] >> (kz[zi] & 31);
            }
        }
        return (any & 1u) != 0;
    }
    for (int xi = 0; xi <= xNum; ++xi)
        for (int yi = 0; yi <= yNum; ++yi)
            for (int zi = 0; zi <= zNum; ++zi) any |= grid_bits_at(g, kx[xi], ky[yi], kz[zi]) >> 1;  // unknown | occupied
    return any != 0;
}

// polyTrajOctomap::checkCollision(point3d) for one pose.  T != nullptr: look the voxels up in the LDS
// tile when it holds them (it does by construction of the tile; the test costs six compares per pose).
__device__ __forceinline__ bool box_sweep(const GridView& g, const SweepConst& C, float fx, float fy, float fz,
                                          const Tile* T, const uint32_t* tile_words, SweepMemo* memo = nullptr,
                                          const CountConst* N = nullptr) {
    const double map_res = C.map_res, rf = C.rf;
    // PO.cpp:548-555
    const double xmin = fx - C.box[0] / 2, xmax = fx + C.box[0] / 2;
    const double ymin = fy - C.box[1] / 2, ymax = fy + C.box[1] / 2;
    const double zmin = fz - C.box[2] / 2, zmax = fz + C.box[2] / 2;
    // truncation of a ROUNDED quotient whose dividend wobbles by an ulp of fx around the box size: the
    // count can be one short, per pose (replacing the divisions by compares against host-bisected
    // thresholds gives the same integers but measured 5 % slower)
    // A pose at infinity or NaN makes the quotient NaN (inf - inf).  The conversion is undefined in C++; the reference as
    // it runs on x86 — and the oracle on this host — gets cvttsd2si's INT_MIN: no pass of the loops, the pose does NOT
    // collide.  The device conversion would return 0 (one lattice point, outside the bounds: collides), hence the select.
    int xNum, yNum, zNum;
    if (N) {                                                        // (wave-uniform)
        const double dx = xmax - xmin, dy = ymax - ymin, dz = zmax - zmin;
        xNum = dx == dx ? (dx >= N->thr[0] ? N->nhi[0] : N->nlo[0]) : (int)0x80000000;
        yNum = dy == dy ? (dy >= N->thr[1] ? N->nhi[1] : N->nlo[1]) : (int)0x80000000;
        zNum = dz == dz ? (dz >= N->thr[2] ? N->nhi[2] : N->nlo[2]) : (int)0x80000000;
    } else {
        const double qxn = (xmax - xmin) / map_res, qyn = (ymax - ymin) / map_res, qzn = (zmax - zmin) / map_res;
        xNum = qxn == qxn ? (int)qxn : (int)0x80000000;
        yNum = qyn == qyn ? (int)qyn : (int)0x80000000;
        zNum = qzn == qzn ? (int)qzn : (int)0x80000000;
    }
    bool hit = false;
    if (xNum < kAxisMax && yNum < kAxisMax && zNum < kAxisMax && xNum >= 0 && yNum >= 0 && zNum >= 0) {
        int kx[kAxisMax], ky[kAxisMax], kz[kAxisMax];
        axis_keys(xmin, xNum, map_res, rf, g.bmin[0], g.bmax[0], g.key0[0], g.nx, kx, hit);
        axis_keys(ymin, yNum, map_res, rf, g.bmin[1], g.bmax[1], g.key0[1], g.ny, ky, hit);
        axis_keys(zmin, zNum, map_res, rf, g.bmin[2], g.bmax[2], g.key0[2], g.nz, kz, hit);
        if (hit) return true;
        if (memo) {
            const int nums = xNum | (yNum << 8) | (zNum << 16);
            bool same = nums == memo->nums;
#pragma unroll
            for (int i = 0; i < kAxisMax; ++i) same = same && kx[i] == memo->kx[i] && ky[i] == memo->ky[i] && kz[i] == memo->kz[i];
            if (__all(same)) return memo->verdict;
            memo->nums = nums;
#pragma unroll
            for (int i = 0; i < kAxisMax; ++i) { memo->kx[i] = kx[i]; memo->ky[i] = ky[i]; memo->kz[i] = kz[i]; }
        }
        const bool v = lattice_any(g, T, tile_words, kx, ky, kz, xNum, yNum, zNum);
        if (memo) memo->verdict = v;
        return v;
    }
    // a collision box of more than 3 map cells per axis: the reference's walk as written
    for (int xi = 0; xi <= xNum && !hit; ++xi) {
        const float qx = (float)(xmin + xi * map_res);
        const bool x_out = !(qx >= g.bmin[0] && qx <= g.bmax[0]);
        const int kx = (int)floor(rf * (double)qx) - g.key0[0];
        for (int yi = 0; yi <= yNum && !hit; ++yi) {
            const float qy = (float)(ymin + yi * map_res);
            const bool y_out = !(qy >= g.bmin[1] && qy <= g.bmax[1]);
            const int ky = (int)floor(rf * (double)qy) - g.key0[1];
            for (int zi = 0; zi <= zNum; ++zi) {
                const float qz = (float)(zmin + zi * map_res);
                if (x_out || y_out || !(qz >= g.bmin[2] && qz <= g.bmax[2])) { hit = true; break; }
                const int kz = (int)floor(rf * (double)qz) - g.key0[2];
                if ((grid_bits_at(g, kx, ky, kz) >> 1) != 0) { hit = true; break; }  // outside -> 7 -> collides
            }
        }
    }
    return hit;
}

// The box lattice's OR for EVERY choice of lattice counts at once (k_corridor's span certificates): bit b of the result,
// b = bx | by << 1 | bz << 2, is the OR over the lattice with count nhi on the axes whose bit is set and nlo on the
// others (nhi - nlo <= 1).  A lattice point beyond nlo on an axis belongs to the choices with that axis' bit set only,
// so one pass over the nhi lattice serves all eight: each lookup ORs its bit into the choices that contain the point.
// (An axis whose count cannot vary has nlo == nhi: both values of its bit get the same answer.)
__device__ __forceinline__ unsigned lattice_any_all(const GridView& g, const Tile* T, const uint32_t* tile_words, const int (&kx)[kAxisMax],
                                                    const int (&ky)[kAxisMax], const int (&kz)[kAxisMax], const CountConst& N) {
    const int xNum = N.nhi[0], yNum = N.nhi[1], zNum = N.nhi[2];
    const bool tiled = T->in_lds && kx[0] >= T->x0 && kx[xNum] < T->x0 + T->tx && ky[0] >= T->y0 &&
                       ky[yNum] < T->y0 + T->ty && (kz[0] >> 5) >= T->w0 && (kz[zNum] >> 5) < T->w0 + T->tw;
    unsigned tt = 0u;
    if (!tiled) {                      // (a tile too large for the LDS: rare, kept small)
#pragma unroll 1
        for (int i = 0; i < kAxisMax * kAxisMax * kAxisMax; ++i) {
            const int xi = i >> 4, yi = (i >> 2) & 3, zi = i & 3;
            if (xi > xNum || yi > yNum || zi > zNum) continue;
            const unsigned m = (xi > N.nlo[0] ? 0xaau : 0xffu) & (yi > N.nlo[1] ? 0xccu : 0xffu) & (zi > N.nlo[2] ? 0xf0u : 0xffu);
            if (grid_bits_at(g, kx[xi], ky[yi], kz[zi]) >> 1) tt |= m;
        }
        return tt;
    }
#pragma unroll
    for (int xi = 0; xi < kAxisMax; ++xi) {
        if (xi > xNum) continue;
        const unsigned mx = xi > N.nlo[0] ? 0xaau : 0xffu;
#pragma unroll
        for (int yi = 0; yi < kAxisMax; ++yi) {
            if (yi > yNum) continue;
            const unsigned mxy = mx & (yi > N.nlo[1] ? 0xccu : 0xffu);
            const int col = ((kx[xi] - T->x0) * T->ty + (ky[yi] - T->y0)) * T->tw - T->w0;
#pragma unroll
            for (int zi = 0; zi < kAxisMax; ++zi) {
                if (zi > zNum) continue;
                const unsigned m = mxy & (zi > N.nlo[2] ? 0xf0u : 0xffu);
                const unsigned bit = (tile_words[col + (kz[zi] >> 5)] >> (kz[zi] & 31)) & 1u;
                tt |= (0u - bit) & m;
            }
        }
    }
    return tt;
}

// box_sweep for the first pass of k_corridor: the lattice counts by compare (CountConst: each is nlo or nhi < kAxisMax),
// no memo, no walk for large boxes.  A NaN count (see box_sweep) on any axis: some loop of the reference makes no pass.
__device__ __forceinline__ bool box_sweep_fast(const GridView& g, const SweepConst& C, const CountConst& N, float fx, float fy,
                                               float fz, const Tile* T, const uint32_t* tile_words) {
    const double map_res = C.map_res, rf = C.rf;
    const double xmin = fx - C.box[0] / 2, xmax = fx + C.box[0] / 2;
    const double ymin = fy - C.box[1] / 2, ymax = fy + C.box[1] / 2;
    const double zmin = fz - C.box[2] / 2, zmax = fz + C.box[2] / 2;
    const double dx = xmax - xmin, dy = ymax - ymin, dz = zmax - zmin;
    if (!(dx == dx && dy == dy && dz == dz)) return false;
    const int xNum = dx >= N.thr[0] ? N.nhi[0] : N.nlo[0];
    const int yNum = dy >= N.thr[1] ? N.nhi[1] : N.nlo[1];
    const int zNum = dz >= N.thr[2] ? N.nhi[2] : N.nlo[2];
    bool hit = false;
    int kx[kAxisMax], ky[kAxisMax], kz[kAxisMax];
    axis_keys(xmin, xNum, map_res, rf, g.bmin[0], g.bmax[0], g.key0[0], g.nx, kx, hit);
    axis_keys(ymin, yNum, map_res, rf, g.bmin[1], g.bmax[1], g.key0[1], g.ny, ky, hit);
    axis_keys(zmin, zNum, map_res, rf, g.bmin[2], g.bmax[2], g.key0[2], g.nz, kz, hit);
    if (hit) return true;
    return lattice_any(g, T, tile_words, kx, ky, kz, xNum, yNum, zNum);
}

// ---- certified spans (k_corridor) -----------------------------------------------------------------------------
// Consecutive samples are ~0.1 mm apart against 100 mm voxels: nearly every run of 64 samples sees the same voxels.
// A span of samples [k0, k0 + len) is decided by ONE evaluation when that can be PROVED to give every sample's verdict:
//   interval  every sample's exact-chain position lies within R of the fast form at one clock value t* inside the span:
//             R = 2 E (both evaluations are within E / 2 of the real polynomial, see sampler_error_bound) + L * dt, L >= sup |p'|
//             over the sampled interval (Bernstein coefficients of p' — convex hull — plus 2^-40 sum d |c_d| T^(d-1) for
//             their own rounding), dt >= |t_k - t*| = (half the span) * |delT| + the drift of the accumulated clock from
//             k * delT (<= n u T: one rounding of at most u T per step);
//   floats    conversion to float is monotone: every sample's float lies in [flo, fhi] = [(float)(p - R), (float)(p + R)];
//   keys      every expression of axis_keys() is a monotone function of the pose's float (a sum with a constant, a
//             product by a positive constant, conversions and floor all round monotonically), so a lattice point whose
//             bounds test passes and whose key agrees AT BOTH ENDS has that key for every sample of the span; one that
//             lies beyond the same bound at both ends is outside for every sample;
//   count     the lattice count (int)((xmax - xmin) / map_res) wobbles with the rounding of fx +- box / 2: it lies in
//             [nlo, nhi] computed per segment from |d - box| <= 2^-50 (max |x| + box); the verdict is an OR over the
//             lattice, monotone in the counts, so equal verdicts for (nlo..) and (nhi..) pin it for anything between.
// A span that cannot be certified is cut in four and tried again; the last few samples go through the per-sample path.
// Results are those of the per-sample walk by construction; the proof obligations are the inequalities above.
struct SpanConst {
    double base[3];      // 2 E + L * drift, rounded up
    double lipd[3];      // L * |delT|, rounded up: metres per sample index
    double half[3];      // box / 2
    CountConst N;
};

__device__ __forceinline__ void axis_span(double half, float flo, float fhi, int nlo, int nhi, double map_res, double rf,
                                          double bmin, double bmax, int key0, int dim, int (&k)[kAxisMax], bool& constant,
                                          bool& surely_out) {
    const double a0 = flo - half, a1 = fhi - half;          // xmin of PO.cpp:548 at both ends of the interval
#pragma unroll
    for (int i = 0; i < kAxisMax; ++i) {
        k[i] = 0;
        if (i <= nhi) {
            const float q0 = (float)(a0 + i * map_res), q1 = (float)(a1 + i * map_res);
            const bool both = q0 >= bmin && q0 <= bmax && q1 >= bmin && q1 <= bmax;     // false for NaN
            const int k0 = (int)floor(rf * (double)q0) - key0, k1 = (int)floor(rf * (double)q1) - key0;
            constant = constant && both && k0 == k1 && k0 >= 0 && k0 < dim;
            if (i <= nlo) surely_out = surely_out || q1 < bmin || q0 > bmax || (both && (k1 < 0 || k0 >= dim));
            k[i] = k0;
        }
    }
}

// 0: not certified, 1: every pose with floats in [flo, fhi] is free, 2: every such pose collides, 3: the keys are the same
// for every such pose but the verdict depends on the pose's own lattice counts: *table holds it per choice of counts
__device__ __forceinline__ int certify_span(const GridView& g, const SweepConst& C, const SpanConst& K, const float (&flo)[3],
                                            const float (&fhi)[3], const Tile* T, const uint32_t* tile_words, int* table) {
    int kx[kAxisMax], ky[kAxisMax], kz[kAxisMax];
    bool constant = true, out = false;
    axis_span(K.half[0], flo[0], fhi[0], K.N.nlo[0], K.N.nhi[0], C.map_res, C.rf, g.bmin[0], g.bmax[0], g.key0[0], g.nx, kx, constant, out);
    axis_span(K.half[1], flo[1], fhi[1], K.N.nlo[1], K.N.nhi[1], C.map_res, C.rf, g.bmin[1], g.bmax[1], g.key0[1], g.ny, ky, constant, out);
    axis_span(K.half[2], flo[2], fhi[2], K.N.nlo[2], K.N.nhi[2], C.map_res, C.rf, g.bmin[2], g.bmax[2], g.key0[2], g.nz, kz, constant, out);
    if (out) {
        // ... provided every float of the span is finite: a pose at infinity on ANY axis has no lattice at all (see
        // box_sweep) and does not collide, whatever this axis says.  (`constant` implies it: the bounds tests passed.)
        bool finite = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) finite = finite && fabsf(flo[a]) <= 3.402823466e38f && fabsf(fhi[a]) <= 3.402823466e38f;   // false for NaN
        return finite ? 2 : 0;
    }
    if (!constant) return 0;
    const unsigned tt = lattice_any_all(g, T, tile_words, kx, ky, kz, K.N);
    if (tt == 0u) return 1;
    if (tt == 0xffu) return 2;          // (the OR is monotone in the counts: the all-nlo choice collides, so does every other)
    *table = (int)tt;
    return 3;
}

constexpr int kItemCap = 4 * kBlock;   // pieces waiting for their certificate, per level (a full queue marks samples instead)
constexpr int kBitWords = 2 * kBlock;  // a batch: up to kBlock spans of up to 64 samples
constexpr int kTtBytes = 16 * kBlock;  // the smallest pieces are a sixteenth of a span
constexpr int kParallelMax = 2 * kBlock;

// The clock tables of all segments, a thread each: the table is thread-serial work (~9 us), which a workgroup of
// k_corridor would otherwise wait for with 255 lanes idle.
__global__ void __launch_bounds__(64) k_corridor_clocks(int S, const int32_t* __restrict__ n_samp, const double* __restrict__ delT,
                                                        ClockTable* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    out[s].n = -1;
    const int n = n_samp[s];
    if (n > 0) (void)build_clock_table(delT[s], n - 1, out[s]);
}

// Two passes, one launch each (PASS 0 then PASS 1), so that neither carries the other's registers:
//   PASS 0  segments of more than 512 samples by certified spans, shorter ones a sample per lane (both need the clock
//           table).  A segment it cannot take — degenerate delT, a box of more than 3 map cells per axis, non-finite
//           coefficients, samples further apart than 1/32 of a voxel, an exact-power queue that overflowed — is left to
//           PASS 1 through A.todo[s];
//   PASS 1  the walk of rounds 1-2 for those: every thread over chunks of 16 consecutive samples.
// DEG7: the planner's degree (cfg polynomial_degree: 7), unrolled.
// (four waves per SIMD = at most 128 VGPRs)
#ifndef VIGO_CORRIDOR_WPS
#define VIGO_CORRIDOR_WPS 4
#endif
template <int PASS, bool DEG7>
__global__ void __launch_bounds__(kBlock, VIGO_CORRIDOR_WPS) k_corridor(GridView g, CorridorArgs A) {
    extern __shared__ __align__(16) uint32_t tile_words[];
    __shared__ double cf[3 * (kMaxDeg + 1)];
    __shared__ int s_min[3], s_max[3];
    __shared__ double s_err[3];
    __shared__ int s_first, s_count;
    __shared__ int q_n, q_idx[kQueueCap];          // samples the float filter could not certify
    __shared__ ClockTable s_clock;
    __shared__ double s_bern[3][kMaxDeg + 1], s_dbern[3][kMaxDeg + 1];
    __shared__ SpanConst s_span;
    __shared__ int s_span_ok[3];
    __shared__ int s_items[2][kItemCap], s_in[2];    // pieces of spans that were not certified, two levels of cutting
    __shared__ uint32_t s_bits[kBitWords];           // one bit per sample of a batch: goes through the per-sample path
    __shared__ uint32_t s_fbits[kBitWords];          // ... : the verdict is s_tt at the pose's own lattice counts
    __shared__ uint8_t s_tt[kTtBytes];               // per smallest piece: verdict for each of the 8 choices of counts

    const int s = blockIdx.x;
    if (s >= A.S) return;
    if (PASS == 1 && !A.todo[s]) return;
    const int tid = threadIdx.x;
    const int deg = DEG7 ? 7 : A.deg;
    const int n = A.n_samp[s];
    const double dT = A.delT[s];

    if (tid < 3 * (deg + 1)) {
        const int ax = tid / (deg + 1), d = tid % (deg + 1);
        cf[ax * (kMaxDeg + 1) + d] = A.coeffs[((size_t)s * 3 + ax) * (deg + 1) + d];
    }
    if (tid < 3) { s_min[tid] = 0x7fffffff; s_max[tid] = (int)0x80000000; s_err[tid] = 0.0; s_span_ok[tid] = 0; }
    if (tid == 0) {
        s_first = 0x7fffffff; s_count = 0; q_n = 0;
        s_clock.n = -1;
    }
    __syncthreads();

    const int n_chunks = (n + kChunk - 1) / kChunk;

    // ---- per-segment constants, spread over the block ----
    //   clock   the sample clock as a table (vigo_exact_time.hpp): copied from k_corridor_clocks' output, or written down
    //           here by one lane of wave 3 (a launch of more segments than the workspace is made for);
    //   bounds  the Bernstein coefficients of the segment and of its derivative over [0, Tu], one per thread of waves 0
    //           and 1 (convex-hull property: min b_i <= p(t) <= max b_i), Tu >= every clock value — the accumulated
    //           clock stays within n u of k delT, u = 2^-53.  The position bound only sizes the LDS tile: a pose whose
    //           lattice points fall outside the tile takes the L2 path in box_sweep, so results never depend on it.  The
    //           derivative bound is the Lipschitz constant of the span certificates (SpanConst).
    const double Tu = n > 0 ? (double)(n - 1) * dT * (1.0 + 0x1p-20) : 0.0;
    if (A.clocks) {
        const int* src = reinterpret_cast<const int*>(A.clocks + s);
        int* dst = reinterpret_cast<int*>(&s_clock);
        for (int i = tid; i < (int)(sizeof(ClockTable) / sizeof(int)); i += kBlock) dst[i] = src[i];
    } else if (tid == 3 * 64 && n > 0) {
        (void)build_clock_table(dT, n - 1, s_clock);
    }
    if (tid < 3 * (deg + 1)) {
        const int a = tid / (deg + 1), i = tid % (deg + 1);
        const double* c = cf + a * (kMaxDeg + 1);
        // b_i = sum_{k <= i} C(i,k) / C(deg,k) * c_k * Tu^k
        double bi = c[0], ratio = 1.0, pw = 1.0;
        for (int k = 1; k <= i; ++k) {
            ratio *= (double)(i - k + 1) / (double)(deg - k + 1);
            pw *= Tu;
            bi += ratio * c[k] * pw;
        }
        s_bern[a][i] = bi;
    }
    if (tid >= 64 && tid < 64 + 3 * deg) {
        const int a = (tid - 64) / deg, i = (tid - 64) % deg;
        const double* c = cf + a * (kMaxDeg + 1);
        // the same for p' (coefficients (k + 1) c_{k+1}, degree deg - 1)
        double bi = c[1], ratio = 1.0, pw = 1.0;
        for (int k = 1; k <= i; ++k) {
            ratio *= (double)(i - k + 1) / (double)(deg - k);
            pw *= Tu;
            bi += ratio * ((double)(k + 1) * c[k + 1]) * pw;
        }
        s_dbern[a][i] = fabs(bi);
    }
    __syncthreads();
    if (tid < 3 && n > 0) {
        const double* c = cf + tid * (kMaxDeg + 1);
        s_err[tid] = sampler_error_bound(c, deg, Tu);         // filter of sample_f32()
        double lo = c[0], hi = c[0];                           // b_0 = c_0
        for (int i = 1; i <= deg; ++i) {
            lo = fmin(lo, s_bern[tid][i]);
            hi = fmax(hi, s_bern[tid][i]);
        }
        const double pad = 1e-6 * (1.0 + fmax(fabs(lo), fabs(hi)));    // rounding of the conversion and of (float)p
        if (lo <= hi) {                                        // false for NaN coefficients
            s_min[tid] = f2ord((float)(lo - pad) - 1e-6f);
            s_max[tid] = f2ord((float)(hi + pad) + 1e-6f);
        }
        // -- span certificates: Lipschitz constant of this axis, size of the positions, range of the lattice count
        const double Tm = fabs(Tu);
        double Ac = 0.0, A1 = 0.0, pwT = 1.0;                  // sum |c_d| T^d,  sum d |c_d| T^(d-1)
        for (int d = 0; d <= deg; ++d) {
            Ac += fabs(c[d]) * pwT;
            if (d < deg) A1 += (double)(d + 1) * fabs(c[d + 1]) * pwT;
            pwT *= Tm;
        }
        double Lb = 0.0;
        for (int i = 0; i < deg; ++i) Lb = fmax(Lb, s_dbern[tid][i]);
        const double L = Lb + 0x1p-40 * A1;                    // (rounding of the coefficients; NaN / inf: A1 carries them)
        const double up = 1.0 + 0x1p-40;
        const double drift = (double)n * 0x1p-52 * (Tm + fabs(dT));     // |t_k - fl(k delT)| for every k < n
        const double base = (2.0 * s_err[tid] + L * drift) * up;
        const double lipd = L * fabs(dT) * up;
        const double h = A.sweep.box[tid] / 2;
        const double Mx = Ac * (1.0 + 0x1p-20) + fabs(h);               // >= |(float)x| + |box / 2| for every sample
        const double dl = 0x1p-50 * (Mx + fabs(h));                     // |(fx + h) - (fx - h) - box| as computed
        double ql = (A.sweep.box[tid] - dl) / A.sweep.map_res, qh = (A.sweep.box[tid] + dl) / A.sweep.map_res;
        ql -= fabs(ql) * 0x1p-50;
        qh += fabs(qh) * 0x1p-50;
        const bool ok = ql > -1.0 && qh < (double)kAxisMax && base < 1e300 && lipd < 1e300;   // false for NaN
        s_span.base[tid] = base;
        s_span.lipd[tid] = lipd;
        s_span.half[tid] = h;
        const int nlo = ok ? (int)ql : 0, nhi = ok ? (int)qh : 0;
        // the dividing line of CountConst: the smallest d with (int)(d / map_res) >= nhi, a few ulps from nhi * map_res
        double thr = -1.0;                                    // nlo == nhi: every difference reaches it
        bool thr_ok = ok && nhi - nlo <= 1;
        if (thr_ok && nhi != nlo) {
            auto cnt = [&](double d) { return (int)(d / A.sweep.map_res); };
            auto step = [](double d, int by) { return __longlong_as_double(__double_as_longlong(d) + by); };   // d > 0
            double c = (double)nhi * A.sweep.map_res;
            int guard = 0;
            while (guard < 8 && cnt(step(c, -1)) >= nhi) { c = step(c, -1); ++guard; }
            while (guard < 16 && cnt(c) < nhi) { c = step(c, 1); ++guard; }
            thr_ok = cnt(c) >= nhi && cnt(step(c, -1)) < nhi;
            thr = c;
        }
        s_span.N.nlo[tid] = nlo;
        s_span.N.nhi[tid] = nhi;
        s_span.N.thr[tid] = thr;
        s_span_ok[tid] = ok ? (thr_ok ? 3 : 1) : 0;           // bit 0: span certificates, bit 1: counts by compare
    }
    __syncthreads();

    // ---- tile of voxels the sweep can touch (uniform across the block) ----
    Tile T;
    {
        int lo_i[3], hi_i[3];
        const int dims[3] = {g.nx, g.ny, g.nz};
        bool any = s_min[0] != 0x7fffffff;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double lo = (double)ord2f(s_min[a]) - A.sweep.box[a] / 2;
            const double hi = (double)ord2f(s_max[a]) + A.sweep.box[a] / 2 + A.sweep.map_res;
            double l = floor(A.sweep.rf * lo) - g.key0[a] - 1;
            double h = floor(A.sweep.rf * hi) - g.key0[a] + 1;
            l = fmax(l, 0.0);
            h = fmin(h, (double)(dims[a] - 1));
            lo_i[a] = (int)l;
            hi_i[a] = (int)h;
            if (!(h >= l)) any = false;
        }
        T.x0 = lo_i[0]; T.y0 = lo_i[1]; T.w0 = lo_i[2] >> 5;
        T.tx = any ? hi_i[0] - lo_i[0] + 1 : 0;
        T.ty = any ? hi_i[1] - lo_i[1] + 1 : 0;
        T.tw = any ? (hi_i[2] >> 5) - T.w0 + 1 : 0;
        const long long words = (long long)T.tx * T.ty * T.tw;
        T.in_lds = any && words > 0 && words <= A.tile_words_cap;
        if (T.in_lds) {
            const uint32_t* unk = g.planes + g.plane_words;
            const uint32_t* occ = g.planes + 2 * g.plane_words;
            // a thread per (x, y) column, its z words in turn: one integer division per column instead of three per word
            const int cols = T.tx * T.ty;
            for (int col = tid; col < cols; col += kBlock) {
                const int lx = col / T.ty, ly = col - lx * T.ty;
                const size_t gw = ((size_t)(T.x0 + lx) * g.ny + (T.y0 + ly)) * g.nzw + T.w0;
                for (int lw = 0; lw < T.tw; ++lw) tile_words[col * T.tw + lw] = unk[gw + lw] | occ[gw + lw];
            }
        }
    }
    __syncthreads();

    // ---- how the samples are visited (block-uniform) ----
    // PASS 0 takes a segment when it has a clock table and its lattice counts go by compare (CountConst):
    //   certify  n > 512 and samples closer than 1/32 of a voxel: certified spans of 64, 32 or 16 samples (the largest whose
    //            reach stays within a quarter of a voxel), cut in four where the certificate fails (see SpanConst);
    //   else     n <= 512: every sample through the per-sample path, a lane each.
    // Everything else — degenerate delT, a box of more than 3 map cells per axis, non-finite coefficients, fast or very
    // long segments, an exact-power queue that overflows — is PASS 1's: the walk of rounds 1-2, every thread over chunks
    // of 16 consecutive samples.
    const bool table = s_clock.n > 0;
    const bool counts = (s_span_ok[0] & s_span_ok[1] & s_span_ok[2] & 2) != 0;
    int S1 = 0;
    if (n > kParallelMax && n <= (1 << 24)) {
        const double lipmax = fmax(s_span.lipd[0], fmax(s_span.lipd[1], s_span.lipd[2]));
        const double cell = 0.25 / A.sweep.rf;                // a quarter of a voxel: the reach of a span's certificate
        S1 = lipmax * 32.0 <= cell ? 64 : lipmax * 16.0 <= cell ? 32 : lipmax * 8.0 <= cell ? 16 : 0;
    }
    const bool certify = S1 > 0;
    if (PASS == 0) {
        const bool mine = table && counts && (certify || n <= kParallelMax);
        if (tid == 0) A.todo[s] = mine ? 0 : 1;
        if (!mine) return;
        if (!certify) S1 = 32;
    }

    int my_first = 0x7fffffff, my_count = 0;
    // (block-uniform: kept in SGPRs — six VGPRs more would cost the kernel its fourth wave per SIMD)
    const double E[3] = {uniform_f64(s_err[0]), uniform_f64(s_err[1]), uniform_f64(s_err[2])};
    {
        // PASS 1 keeps the coefficients of the planner's degree in registers; PASS 0 reads them from LDS
        constexpr bool REG = PASS == 1;
        double c7[3][8];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int d = 0; d < 8; ++d) c7[a][d] = (DEG7 && REG) ? cf[a * (kMaxDeg + 1) + d] : 0.0;
        // the lattice counts by compare where the segment's constants allow it (CountConst), block-uniform
        CountConst Nc;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            Nc.nlo[a] = __builtin_amdgcn_readfirstlane(s_span.N.nlo[a]);
            Nc.nhi[a] = __builtin_amdgcn_readfirstlane(s_span.N.nhi[a]);
            Nc.thr[a] = uniform_f64(s_span.N.thr[a]);
        }
        if constexpr (PASS == 0) {
            SpanConst K;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                K.base[a] = uniform_f64(s_span.base[a]);
                K.lipd[a] = uniform_f64(s_span.lipd[a]);
                K.half[a] = uniform_f64(s_span.half[a]);
            }
            K.N = Nc;
            // Batches of up to kBlock spans of S1 samples (64, 32 or 16), three rounds of certificates and the rest:
            //   1  a lane per span: certificate, else its four quarters queued;
            //   2  a lane per queued quarter: certificate, else ITS quarters queued;
            //   3  a lane per queued sixteenth: certificate, else its samples marked in s_bits;
            //      (a piece of 2 samples or fewer is not queued but marked; so is one that finds its queue full.  A piece
            //      whose keys are constant but whose verdict hangs on each pose's own lattice counts is marked in
            //      s_fbits, with the verdict per choice of counts in s_tt)
            //   4  the marked samples, compacted per wave (prefix sums over the words' popcounts), a lane per sample:
            //      s_bits through the whole per-sample path, s_fbits through the sampler, the three counts and s_tt.
            const int wave = tid >> 6, lane = tid & 63;
            const int g_shift = S1 == 64 ? 2 : S1 == 32 ? 1 : 2;       // log2 of the smallest piece: 4, 2, 4 samples
            int k_base = 0;
            auto mark_in = [&](uint32_t* bits, int k0, int len) {
                int b = k0 - k_base;
                while (len > 0) {
                    const int take = min(len, 32 - (b & 31));
                    atomicOr(&bits[b >> 5], (take >= 32 ? 0xffffffffu : ((1u << take) - 1u)) << (b & 31));
                    b += take;
                    len -= take;
                }
            };
            // certificate of one piece: 1 decided and accounted for, 3 marked in s_fbits, 0 open
            auto decide = [&](int k0, int len) -> int {
                const int c = k0 + (len >> 1);
                const int hs = max(c - k0, k0 + len - 1 - c);
                const double ts = fmin(fmax((double)c * dT, 0.0), Tu);   // a clock value within reach of the piece (delT > 0 here)
                double p[3];
                if (DEG7) poly_fast7_lds(cf, ts, p);
                else poly_fast(cf, deg, ts, p);
                float flo[3], fhi[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double R = (K.base[a] + K.lipd[a] * (double)hs) * (1.0 + 0x1p-40);
                    flo[a] = (float)(p[a] - R);
                    fhi[a] = (float)(p[a] + R);
                }
                int tt = 0;
                const int v = certify_span(g, A.sweep, K, flo, fhi, &T, tile_words, &tt);
                if (v == 2) {
                    if (k0 < my_first) my_first = k0;
                    my_count += len;
                }
                if (v == 3) {
                    mark_in(s_fbits, k0, len);
                    for (int o = 0; o < len; o += 1 << g_shift) s_tt[(k0 + o - k_base) >> g_shift] = (uint8_t)tt;
                }
                return v == 2 ? 1 : v;
            };
            const int n_spans = (n + S1 - 1) / S1;
            const int n_batches = (n_spans + kBlock - 1) / kBlock;
            const int per_batch = (n_spans + n_batches - 1) / n_batches;  // <= kBlock spans, the batches alike
            const int n_words = (per_batch * S1 + 31) >> 5;              // <= kBitWords
            for (int base = 0; base < n_spans; base += per_batch) {     // block-uniform trip count
                k_base = base * S1;
                for (int w = tid; w < n_words; w += kBlock) { s_bits[w] = 0u; s_fbits[w] = 0u; }
                if (tid < 2) s_in[tid] = 0;
                __syncthreads();
#pragma unroll 1
                for (int phase = 0; phase < 3; ++phase) {
                    const int cnt = phase == 0 ? min(per_batch, n_spans - base) : min(s_in[(phase - 1) & 1], kItemCap);
                    const int child = (S1 >> 2) >> (2 * phase);             // a quarter of this round's pieces
                    // (consecutive pieces on consecutive waves instead of consecutive lanes: measured 2 % slower)
                    for (int i = tid; i < cnt; i += kBlock) {
                        int k0, len;
                        if (phase == 0) {
                            k0 = (base + i) * S1;
                            len = min(S1, n - k0);
                        } else {
                            const int it = s_items[(phase - 1) & 1][i];
                            k0 = it >> 7;
                            len = it & 127;
                        }
                        if (!certify) mark_in(s_bits, k0, len);
                        else if (decide(k0, len) == 0) {
                            if (phase == 2 || child < 2 || len <= child) mark_in(s_bits, k0, len);
                            else {
                                for (int o = 0; o < len; o += child) {
                                    const int l = min(child, len - o);
                                    const int slot = l > 2 ? atomicAdd(&s_in[phase & 1], 1) : kItemCap;
                                    if (slot < kItemCap) s_items[phase & 1][slot] = ((k0 + o) << 7) | l;
                                    else mark_in(s_bits, k0 + o, l);
                                }
                            }
                        }
                    }
                    __syncthreads();
                }
                // the marked samples of both bit maps, a lane each: in round r lane j of wave w takes word kBlock r + 4 j + w,
                // so that every wave sees the whole batch at a stride of four words
#pragma unroll 1
                for (int which = 0; which < 2; ++which) {
                    const uint32_t* bits = which ? s_fbits : s_bits;
#pragma unroll 1
                    for (int w0 = 0; w0 < n_words; w0 += kBlock) {
                        const int my_word = w0 + lane * 4 + wave;
                        const uint32_t W = my_word < n_words ? bits[my_word] : 0u;
                        const int cw = __popc(W);
                        // (the clock table's piece at the word's first sample, looked up once per word: its samples start there)
                        const int piece_w = W ? clock_piece(s_clock, k_base + (my_word << 5)) : 0;
                        int incl = cw;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const int up = __shfl_up(incl, d);
                            if (lane >= d) incl += up;
                        }
                        const int total = __shfl(incl, 63), excl = incl - cw;
#pragma unroll 1
                        for (int r0 = 0; r0 < total; r0 += 64) {
                            const int r = r0 + lane;
                            int L = 0;                                       // the first lane whose inclusive count exceeds r
#pragma unroll
                            for (int st = 32; st >= 1; st >>= 1)
                                if (__shfl(incl, L + st - 1) <= r) L += st;
                            L = min(L, 63);
                            const uint32_t WL = __shfl(W, L);
                            int rr = r - __shfl(excl, L), pos = 0;          // the rr-th set bit of that lane's word
#pragma unroll
                            for (int st = 16; st >= 1; st >>= 1) {
                                const int below = __popc((WL >> pos) & ((1u << st) - 1u));
                                if (rr >= below) { rr -= below; pos += st; }
                            }
                            const int piece = __shfl(piece_w, L);
                            if (r < total) {
                                const int k = k_base + ((w0 + L * 4 + wave) << 5) + pos;
                                const double t = clock_from(s_clock, piece, k);
                                float f[3];
                                if (sample_f32_fast<DEG7, false>(c7, cf, deg, t, E, f)) {    // pose2Octomap of getPose(t), certified
                                    bool hit;
                                    if (which == 0) hit = box_sweep_fast(g, A.sweep, Nc, f[0], f[1], f[2], &T, tile_words);
                                    else {
                                        // the three lattice counts of this pose (box_sweep's own expressions) pick the verdict
                                        int idx = 0;
#pragma unroll
                                        for (int a = 0; a < 3; ++a) {
                                            const double lo = f[a] - K.half[a], hi = f[a] + K.half[a];
                                            if ((hi - lo) >= Nc.thr[a]) idx |= 1 << a;
                                        }
                                        hit = ((s_tt[(k - k_base) >> g_shift] >> idx) & 1) != 0;
                                    }
                                    if (hit) {
                                        if (k < my_first) my_first = k;
                                        ++my_count;
                                    }
                                } else {
                                    const int slot = atomicAdd(&q_n, 1);
                                    if (slot < kQueueCap) q_idx[slot] = k;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
            if (q_n > kQueueCap) {          // the exact-power queue overflowed: the second pass starts over
                if (tid == 0) A.todo[s] = 1;
                return;
            }
        } else {
            const CountConst* Np = counts ? &Nc : nullptr;
            // every lane remembers the keys and verdict of its previous pose (SweepMemo)
            SweepMemo memo;
            memo.nums = -1;
            memo.verdict = false;
#pragma unroll
            for (int i = 0; i < kAxisMax; ++i) memo.kx[i] = memo.ky[i] = memo.kz[i] = 0;
            for (int c = tid; c < n_chunks; c += kBlock) {
                const int k0 = c * kChunk, k1 = min(n, k0 + kChunk);
                double t = table ? clock_at(s_clock, k0) : accumulated_time(dT, k0);
                for (int k = k0; k < k1; ++k) {
                    float f[3];
                    if (sample_f32_fast<DEG7, true>(c7, cf, deg, t, E, f)) {    // pose2Octomap of getPose(t), certified
                        if (box_sweep(g, A.sweep, f[0], f[1], f[2], &T, tile_words, &memo, Np)) {
                            if (k < my_first) my_first = k;
                            ++my_count;
                        }
                    } else {
                        const int slot = atomicAdd(&q_n, 1);               // (q_n counts past the capacity: see below)
                        if (slot < kQueueCap) q_idx[slot] = k;
                    }
                    t += dT;
                }
            }
        }
    }
    __syncthreads();
    // ---- the queued samples, with the exact-power chain.  A queue that overflowed (non-finite coefficients, a
    //      polynomial that cancels to ~0 over its whole span) is replaced by a walk over all samples that repeats
    //      the filter and handles exactly those it rejects — the same set.  (PASS 1 only: PASS 0 has handed such a
    //      segment over.) ----
    {
        const int queued = q_n;
        const bool all = PASS == 1 && queued > kQueueCap;
        const int count = all ? n : queued;
        const double c0[3][8] = {};
        for (int i = tid; i < count; i += kBlock) {
            const int k = all ? i : q_idx[i];
            const double t = accumulated_time(dT, k);
            float f[3];
            if (all && sample_f32_fast<false>(c0, cf, deg, t, E, f)) continue;
            sample_f32_exact(cf, deg, t, f);
            if (box_sweep(g, A.sweep, f[0], f[1], f[2], &T, tile_words, nullptr)) {
                if (k < my_first) my_first = k;
                ++my_count;
            }
        }
    }
    if (my_count) {
        atomicMin(&s_first, my_first);
        atomicAdd(&s_count, my_count);
    }
    __syncthreads();
    if (tid == 0) {
        A.out_flag[s] = (uint8_t)(s_count > 0);
        if (A.out_first) A.out_first[s] = s_count > 0 ? s_first : -1;
        if (A.out_count) A.out_count[s] = s_count;
    }
}

// ---- the sampler alone: polyTrajSolver::getTrajectory (PS.cpp:1125-1137) for S segments ---------------
// Same clock (accumulated_time + the reference's t += delT inside a chunk) and the same poly_pos / poly_pos7 as
// k_corridor, so a parity test of these positions is a parity test of what the checker sweeps.
__global__ void __launch_bounds__(kBlock) k_poly_sample(int S, int deg, const double* __restrict__ coeffs,
                                                        const int32_t* __restrict__ n_samp, const double* __restrict__ delT,
                                                        int stride, double* __restrict__ out_pos, float* __restrict__ out_f32) {
    __shared__ double cf[3 * (kMaxDeg + 1)];
    const int s = blockIdx.x;
    if (s >= S) return;
    const int tid = threadIdx.x;
    const int n = min(n_samp[s], stride);
    const double dT = delT[s];
    if (tid < 3 * (deg + 1)) {
        const int ax = tid / (deg + 1), d = tid % (deg + 1);
        cf[ax * (kMaxDeg + 1) + d] = coeffs[((size_t)s * 3 + ax) * (deg + 1) + d];
    }
    __syncthreads();
    double c7[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 8; ++d) c7[a][d] = deg == 7 ? cf[a * (kMaxDeg + 1) + d] : 0.0;
    // the float output takes the checker's own route (filtered fast form, sample_f32); the fp64 output is the
    // exact-power chain itself
    double E[3] = {0.0, 0.0, 0.0};
    if (n > 0) {
        const double tl = accumulated_time(dT, n - 1);
#pragma unroll
        for (int a = 0; a < 3; ++a) E[a] = sampler_error_bound(cf + a * (kMaxDeg + 1), deg, tl);
    }
    const int n_chunks = (n + kChunk - 1) / kChunk;
    for (int c = tid; c < n_chunks; c += kBlock) {
        const int k0 = c * kChunk, k1 = min(n, k0 + kChunk);
        double t = accumulated_time(dT, k0);
        for (int k = k0; k < k1; ++k) {
            const size_t o = ((size_t)s * stride + k) * 3;
            if (out_pos) {
                double p[3];
                if (deg == 7) poly_pos7(c7, t, p);
                else poly_pos(cf, deg, t, p);
                out_pos[o] = p[0]; out_pos[o + 1] = p[1]; out_pos[o + 2] = p[2];
            }
            if (out_f32) {
                float f[3];
                const bool ok = deg == 7 ? sample_f32_fast<true>(c7, cf, deg, t, E, f) : sample_f32_fast<false>(c7, cf, deg, t, E, f);
                if (!ok) sample_f32_exact(cf, deg, t, f);
                out_f32[o] = f[0]; out_f32[o + 1] = f[1]; out_f32[o + 2] = f[2];
            }
            t += dT;
        }
    }
}

// ---- box sweep at given sample positions (polyTrajOctomap::checkCollision per pose) -------------
// One thread per pose; lookups go to the packed planes (L2).  Serves the reference's
// checkCollisionTraj(trajectory, ...) signatures, where the samples already exist (PO.cpp:619-656).
__global__ void k_box_points(GridView g, int64_t M, const double* __restrict__ pts, SweepConst C, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float fx = (float)pts[3 * i], fy = (float)pts[3 * i + 1], fz = (float)pts[3 * i + 2];  // pose2Octomap
    const bool hit = box_sweep(g, C, fx, fy, fz, nullptr, nullptr);
    out[i] = (uint8_t)hit;
}

// ---- trilinear ESDF (own definition, see oracle/vigo_oracle.c vgo_esdf_query) ------------
// One query per lane (two or four per lane with all their corner loads in flight measured the same: the gather is
// bound by the lines it touches, not by latency).  The eight corners of a cell are one base address + constants.
__global__ void k_esdf_query(EsdfView E, int64_t Q, const double* __restrict__ pts,
                             double* __restrict__ out_d, double* __restrict__ out_g) {
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its own L2): block b
    // takes the tile (b % 8) * (tiles / 8) + b / 8, so that every XCD walks ONE contiguous eighth of the queries — for
    // spatially coherent queries its L2 then holds lattice lines no other XCD asks for.  (Uniformly random: no effect.)
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned base = nb >> 3, rem = nb & 7u;
    const unsigned tile = xcd * base + (xcd < rem ? xcd : rem) + idx;
    const int64_t q = (int64_t)tile * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int n[3] = {E.nx, E.ny, E.nz};
    int i0[3];
    double f[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double u = (pts[q * 3 + a] - E.origin[a]) / E.res - 0.5;
        const double fl = floor(u);
        int i = (int)fl;
        double fr = u - fl;
        if (i < 0) { i = 0; fr = 0.0; }
        if (i > n[a] - 2) { i = n[a] - 2; fr = 1.0; }
        i0[a] = i;
        f[a] = fr;
    }
    // cell (x, y, z) reads line (x, y / 3, z / 3); its corners are 16 dx + 4 dy + dz further on
    const unsigned by = (unsigned)i0[1] / 3u, bz = (unsigned)i0[2] / 3u;
    const unsigned ly = (unsigned)i0[1] - 3u * by, lz = (unsigned)i0[2] - 3u * bz;
    const float* cell = E.dist + (((size_t)i0[0] * E.nby + by) * E.nbz + bz) * 32 + (ly * 4 + lz);
    double v[2][2][2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            v[dx][dy][0] = (double)cell[dx * 16 + dy * 4];
            v[dx][dy][1] = (double)cell[dx * 16 + dy * 4 + 1];
        }
    const double c00 = v[0][0][0] * (1 - f[0]) + v[1][0][0] * f[0];
    const double c01 = v[0][0][1] * (1 - f[0]) + v[1][0][1] * f[0];
    const double c10 = v[0][1][0] * (1 - f[0]) + v[1][1][0] * f[0];
    const double c11 = v[0][1][1] * (1 - f[0]) + v[1][1][1] * f[0];
    const double c0 = c00 * (1 - f[1]) + c10 * f[1];
    const double c1 = c01 * (1 - f[1]) + c11 * f[1];
    out_d[q] = c0 * (1 - f[2]) + c1 * f[2];
    const double gx00 = v[1][0][0] - v[0][0][0], gx01 = v[1][0][1] - v[0][0][1];
    const double gx10 = v[1][1][0] - v[0][1][0], gx11 = v[1][1][1] - v[0][1][1];
    const double gx0 = gx00 * (1 - f[1]) + gx10 * f[1];
    const double gx1 = gx01 * (1 - f[1]) + gx11 * f[1];
    out_g[q * 3 + 0] = (gx0 * (1 - f[2]) + gx1 * f[2]) / E.res;
    const double gy0 = c10 - c00, gy1 = c11 - c01;
    out_g[q * 3 + 1] = (gy0 * (1 - f[2]) + gy1 * f[2]) / E.res;
    out_g[q * 3 + 2] = (c1 - c0) / E.res;
}

// The same query at the I/O width SURVEY.md §8(d) config 5 states for this fp32 lattice: float3 point in, float value +
// float3 gradient out (12 + 16 B per query instead of 24 + 32), all arithmetic in fp32 on the fp32 corners.  Own
// definition like the fp64 entry (no reference counterpart); each operation rounded once, in this order
// (-ffp-contract=off; oracle/vigo_oracle.c vgo_esdf_query_f32 is the same text):
//   u = (p - (float)origin) * inv_res - 0.5f with inv_res = 1.0f / (float)res computed ONCE on the host,
//   i = floorf(u) clamped to [0, n - 2] (frac 0 / 1 at the clamps), the trilinear blend x then y then z as above,
//   gradient differences * inv_res.  The result is ONE 16-byte store {d, gx, gy, gz}.
struct EsdfF32Const {
    float origin[3];
    float inv_res;
};
__global__ void k_esdf_query_f32(EsdfView E, EsdfF32Const C, int64_t Q, const float* __restrict__ pts, float4* __restrict__ out) {
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;     // XCD-aware tile map, as above
    const unsigned base = nb >> 3, rem = nb & 7u;
    const unsigned tile = xcd * base + (xcd < rem ? xcd : rem) + idx;
    const int64_t q = (int64_t)tile * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int n[3] = {E.nx, E.ny, E.nz};
    int i0[3];
    float f[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = (pts[q * 3 + a] - C.origin[a]) * C.inv_res - 0.5f;
        const float fl = floorf(u);
        // (NaN and out-of-range points: the comparisons are made on the float, so the int conversion is never out of range)
        int i;
        float fr = u - fl;
        if (!(fl >= 0.0f)) { i = 0; fr = 0.0f; }
        else if (fl > (float)(n[a] - 2)) { i = n[a] - 2; fr = 1.0f; }
        else i = (int)fl;
        i0[a] = i;
        f[a] = fr;
    }
    const unsigned by = (unsigned)i0[1] / 3u, bz = (unsigned)i0[2] / 3u;
    const unsigned ly = (unsigned)i0[1] - 3u * by, lz = (unsigned)i0[2] - 3u * bz;
    const float* cell = E.dist + (((size_t)i0[0] * E.nby + by) * E.nbz + bz) * 32 + (ly * 4 + lz);
    float v[2][2][2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            v[dx][dy][0] = cell[dx * 16 + dy * 4];
            v[dx][dy][1] = cell[dx * 16 + dy * 4 + 1];
        }
    const float c00 = v[0][0][0] * (1 - f[0]) + v[1][0][0] * f[0];
    const float c01 = v[0][0][1] * (1 - f[0]) + v[1][0][1] * f[0];
    const float c10 = v[0][1][0] * (1 - f[0]) + v[1][1][0] * f[0];
    const float c11 = v[0][1][1] * (1 - f[0]) + v[1][1][1] * f[0];
    const float c0 = c00 * (1 - f[1]) + c10 * f[1];
    const float c1 = c01 * (1 - f[1]) + c11 * f[1];
    float4 r;
    r.x = c0 * (1 - f[2]) + c1 * f[2];
    const float gx00 = v[1][0][0] - v[0][0][0], gx01 = v[1][0][1] - v[0][0][1];
    const float gx10 = v[1][1][0] - v[0][1][0], gx11 = v[1][1][1] - v[0][1][1];
    const float gx0 = gx00 * (1 - f[1]) + gx10 * f[1];
    const float gx1 = gx01 * (1 - f[1]) + gx11 * f[1];
    r.y = (gx0 * (1 - f[2]) + gx1 * f[2]) * C.inv_res;
    const float gy0 = c10 - c00, gy1 = c11 - c01;
    r.z = (gy0 * (1 - f[2]) + gy1 * f[2]) * C.inv_res;
    r.w = (c1 - c0) * C.inv_res;
    out[q] = r;
}

// one thread per destination float (coalesced writes); values past the lattice edge are zero-filled, never read.
// Line (x, by, bz) holds the values [x, x + 1] x [3 by, 3 by + 3] x [3 bz, 3 bz + 3], z fastest.
__global__ void k_esdf_brick(int nx, int ny, int nz, int nby, int nbz, size_t total, const float* __restrict__ src,
                             float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t line = i >> 5;
    const int r = (int)(i & 31);
    const int bz = (int)(line % nbz), by = (int)((line / nbz) % nby);
    const int x = (int)(line / ((size_t)nbz * nby)) + (r >> 4), y = by * 3 + ((r >> 2) & 3), z = bz * 3 + (r & 3);
    dst[i] = (x < nx && y < ny && z < nz) ? src[((size_t)x * ny + y) * nz + z] : 0.0f;
}

}  // namespace

size_t corridor_clock_ws_bytes(int S) { return (size_t)S * sizeof(ClockTable); }

int launch_esdf_brick(hipStream_t s, int nx, int ny, int nz, const float* src, float* dst) {
    const size_t total = esdf_bricked_floats(nx, ny, nz);
    const int block = 256;
    hipLaunchKernelGGL(k_esdf_brick, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0, s, nx, ny, nz,
                       esdf_bricks_along(ny), esdf_bricks_along(nz), total, src, dst);
    return (int)hipGetLastError();
}

int launch_corridor_check2(hipStream_t s, const GridView& g, int S, int deg, const double* coeffs,
                          const int32_t* n_samp, const double* delT, const double box[3],
                          double map_res, uint8_t* out_flag, int32_t* out_first, int32_t* out_count, int* todo, void* clock_ws) {
    if (S <= 0) return hipSuccess;
    CorridorArgs A{};
    A.S = S; A.deg = deg;
    A.coeffs = coeffs; A.n_samp = n_samp; A.delT = delT;
    A.sweep = SweepConst{{box[0], box[1], box[2]}, map_res, 1.0 / g.res};
    A.out_flag = out_flag; A.out_first = out_first; A.out_count = out_count;
    // with the first pass' static LDS (21.6 KB): VIGO_CORRIDOR_WPS workgroups per CU (18 KB of tile for four)
    const int tile_bytes = (160 * 1024 / VIGO_CORRIDOR_WPS - 22 * 1024) & ~255;
    A.tile_words_cap = tile_bytes / 4;
    A.todo = todo;
    A.clocks = static_cast<const ClockTable*>(clock_ws);
    if (clock_ws)
        hipLaunchKernelGGL(k_corridor_clocks, dim3((S + 63) / 64), dim3(64), 0, s, S, n_samp, delT, static_cast<ClockTable*>(clock_ws));
    if (deg == 7) {
        hipLaunchKernelGGL((k_corridor<0, true>), dim3(S), dim3(kBlock), tile_bytes, s, g, A);
        hipLaunchKernelGGL((k_corridor<1, true>), dim3(S), dim3(kBlock), tile_bytes, s, g, A);
    } else {
        hipLaunchKernelGGL((k_corridor<0, false>), dim3(S), dim3(kBlock), tile_bytes, s, g, A);
        hipLaunchKernelGGL((k_corridor<1, false>), dim3(S), dim3(kBlock), tile_bytes, s, g, A);
    }
    return (int)hipGetLastError();
}

int launch_poly_sample(hipStream_t s, int S, int deg, const double* coeffs, const int32_t* n_samp, const double* delT,
                       int stride, double* out_pos, float* out_f32) {
    if (S <= 0 || stride <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_poly_sample, dim3(S), dim3(kBlock), 0, s, S, deg, coeffs, n_samp, delT, stride, out_pos, out_f32);
    return (int)hipGetLastError();
}

int launch_box_points(hipStream_t s, const GridView& g, int64_t M, const double* pts, const double box[3],
                      double map_res, uint8_t* out) {
    if (M <= 0) return hipSuccess;
    const int block = 256;
    hipLaunchKernelGGL(k_box_points, dim3((unsigned)((M + block - 1) / block)), dim3(block), 0, s, g, M, pts,
                       SweepConst{{box[0], box[1], box[2]}, map_res, 1.0 / g.res}, out);
    return (int)hipGetLastError();
}

int launch_esdf_query(hipStream_t s, const EsdfView& e, int64_t Q, const double* pts,
                      double* out_dist, double* out_grad) {
    if (Q <= 0) return hipSuccess;
    const int block = 256;
    hipLaunchKernelGGL(k_esdf_query, dim3((unsigned)((Q + block - 1) / block)), dim3(block), 0, s, e, Q, pts,
                       out_dist, out_grad);
    return (int)hipGetLastError();
}

int launch_esdf_query_f32(hipStream_t s, const EsdfView& e, int64_t Q, const float* pts, float* out4) {
    if (Q <= 0) return hipSuccess;
    EsdfF32Const C;
    for (int a = 0; a < 3; ++a) C.origin[a] = (float)e.origin[a];
    C.inv_res = 1.0f / (float)e.res;
    const int block = 256;
    hipLaunchKernelGGL(k_esdf_query_f32, dim3((unsigned)((Q + block - 1) / block)), dim3(block), 0, s, e, C, Q, pts,
                       reinterpret_cast<float4*>(out4));
    return (int)hipGetLastError();
}

}  // namespace vigo

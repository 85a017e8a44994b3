// vigo_api.cpp — the C ABI of libvigo_hip.so (include/vigo.h): handle management, argument
// checks and kernel launches.  No computation happens on the host: if HIP is unusable every
// entry point fails loudly (VIGO_ERR_NO_DEVICE / VIGO_ERR_HIP) instead of falling back.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "vigo_exact_pow.hpp"
#include "vigo_corridor.hpp"
#include "vigo_exact_time.hpp"
#include "build/vigo_build_id.h"
#include "vigo_internal.hpp"

using vigo::DevConst;
using vigo::GridView;
using vigo::SolveArgs;

namespace {

int fail(vigo_handle_t h, int code, const char* what, hipError_t e = hipSuccess) {
    if (h) {
        h->last_error = what;
        if (e != hipSuccess) {
            h->last_error += ": ";
            h->last_error += hipGetErrorString(e);
        }
    }
    return code;
}

// (the launchers report hipGetLastError(): whatever another library of the process — torch, RCCL — left
// in the thread's last-error slot is discarded first, so it is not mistaken for a failed launch of ours)
#define VIGO_HIP(h, call)                                                      \
    do {                                                                       \
        (void)hipGetLastError();                                               \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return fail((h), VIGO_ERR_HIP, #call, e_);       \
    } while (0)

int ensure_scratch(vigo_handle_t h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return VIGO_OK;
    if (h->scratch) (void)hipFree(h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    VIGO_HIP(h, hipMalloc(&h->scratch, want));
    h->scratch_bytes = want;
    return VIGO_OK;
}

int check_solve_args(vigo_handle_t h, int B, int N, const void* ctrl) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && !ctrl)) return fail(h, VIGO_ERR_INVALID_ARG, "B < 0 or ctrl == NULL");
    if (N < 7 || N > VIGO_MAX_CTRL_POINTS) return fail(h, VIGO_ERR_UNSUPPORTED_N, "N outside [7, VIGO_MAX_CTRL_POINTS]");
    return VIGO_OK;
}

// The list arguments shared by the solve entry points and the dynamic gate.  Offsets (or a shared count)
// without the list they index mean "no guides" / "no obstacles" — callers keep all-zero CSR offsets around
// when a batch happens to have none — and the kernels never dereference the missing list.
int check_list_args(vigo_handle_t h, const void* guide_off, const void* guide_pv, const void* obs_off, const void* obs, int n_obs_shared) {
    (void)guide_off; (void)guide_pv; (void)obs_off; (void)obs;
    if (n_obs_shared < 0) return fail(h, VIGO_ERR_INVALID_ARG, "n_obs_shared < 0");
    return VIGO_OK;
}

// Sample times of `for (t = 0; t <= tmax; t += dt)` (BT.h:313, :347): t_k is the k-fold floating point
// accumulation, reproduced exactly by vigo::accumulated_time (closed form per binade).  The sample count T is
// found on the host by bisection over that closed form (monotone in k), the table is filled by a device kernel
// and cached in the handle per (dt, tmax): the gates run without a host round trip.
// number of samples of `for (t = 0; t <= tmax; t += dt)` (the accumulated clock, exactly): bisection over the closed form,
// monotone in k.  -1: more than 2^24 samples.
static int count_sample_times(double dt, double tmax) {
    if (!(tmax >= 0.0)) return 0;
    const int64_t cap = (int64_t)1 << 24;
    if (vigo::accumulated_time(dt, cap) <= tmax) return -1;
    int64_t lo = 0, hi = cap;                       // t_lo <= tmax < t_hi
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (vigo::accumulated_time(dt, mid) <= tmax) lo = mid; else hi = mid;
    }
    return (int)hi;                                 // samples k = 0 .. lo
}

int upload_sample_times(vigo_handle_t h, double tmax, double dt, int* out_T, const double** out_dev) {
    if (!(dt > 0.0)) return fail(h, VIGO_ERR_INVALID_ARG, "dt must be > 0");
    if (h->times_T >= 0 && h->times_dt == dt && h->times_tmax == tmax) {
        if (h->times_stream != h->stream) {            // filled on another stream: make sure it has landed
            VIGO_HIP(h, hipStreamSynchronize(h->times_stream));
            h->times_stream = h->stream;
        }
        *out_T = h->times_T;
        *out_dev = h->times_dev;
        return VIGO_OK;
    }
    const int T = count_sample_times(dt, tmax);
    if (T < 0) return fail(h, VIGO_ERR_INVALID_ARG, "too many samples");
    if ((size_t)T > h->times_cap) {
        if (h->times_dev) (void)hipFree(h->times_dev);
        h->times_dev = nullptr;
        h->times_cap = 0;
        h->times_T = -1;
        const size_t want = (size_t)T + 64;
        VIGO_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->times_dev), want * sizeof(double)));
        h->times_cap = want;
    }
    h->times_T = -1;
    VIGO_HIP(h, (hipError_t)vigo::launch_fill_sample_times(h->stream, dt, T, h->times_dev));
    h->times_dt = dt;
    h->times_tmax = tmax;
    h->times_T = T;
    h->times_stream = h->stream;
    *out_T = T;
    *out_dev = h->times_dev;
    return VIGO_OK;
}

// The box sweep visits (box / map_res + 1) lattice points per axis for every pose: a finite box and a bounded
// lattice, or the sweep is an unbounded device loop (the cfg box is 0.4 x 0.4 x 0.2 at 0.2: 3 x 3 x 2 points).
// Negative extents are the reference's "no lattice point at all" and stay allowed.
bool sweep_box_ok(const double box[3], double map_res) {
    double pts = 1.0;
    for (int a = 0; a < 3; ++a) {
        if (!(fabs(box[a]) < 1e9)) return false;
        const double n = box[a] > 0 ? floor(box[a] / map_res) + 1.0 : 1.0;
        if (!(n <= 256.0)) return false;
        pts *= n;
    }
    return pts <= 32768.0;
}

// a finite origin and a finite positive resolution (the key offsets below are integer conversions of origin / res)
bool geometry_ok(const double origin[3], double res) {
    if (!(res > 0.0) || !(res < 1e12)) return false;
    for (int a = 0; a < 3; ++a)
        if (!(fabs(origin[a]) < 1e12) || !(fabs(origin[a] / res) < 2e9)) return false;
    return true;
}

void fill_grid_view(vigo_handle_t h, int nx, int ny, int nz, const double origin[3], double res) {
    GridView& g = h->grid;
    g.planes = h->grid_planes;
    g.nx = nx; g.ny = ny; g.nz = nz;
    g.nzw = (nz + 31) / 32;
    g.plane_words = (size_t)nx * ny * g.nzw;
    g.res = res;
    for (int a = 0; a < 3; ++a) {
        g.origin[a] = origin[a];
        g.bmin[a] = origin[a];
        g.key0[a] = (int)floor(origin[a] / res + 0.5);
    }
    g.bmax[0] = origin[0] + nx * res;
    g.bmax[1] = origin[1] + ny * res;
    g.bmax[2] = origin[2] + nz * res;
    h->has_grid = true;
}

int ensure_grid_storage(vigo_handle_t h, int nx, int ny, int nz) {
    size_t need = vigo_grid_packed_bytes(nx, ny, nz);
    if (need > h->grid_capacity_bytes) {
        if (h->grid_planes) (void)hipFree(h->grid_planes);
        h->grid_planes = nullptr;
        h->grid_capacity_bytes = 0;
        VIGO_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->grid_planes), need));
        h->grid_capacity_bytes = need;
    }
    return VIGO_OK;
}

}  // namespace

extern "C" {

int vigo_abi_version(void) { return 1; }
double vigo_accumulated_time(double delT, int64_t k) { return vigo::accumulated_time(delT, k); }
double vigo_clock_table_time(double delT, int64_t k_last, int64_t k) {
    if (k_last < 0 || k < 0 || k > k_last || k_last > (int64_t)1 << 30) return NAN;
    vigo::ClockTable C;
    const double t_last = vigo::build_clock_table(delT, (int)k_last, C);
    if (C.n <= 0) return NAN;
    if (k == k_last && vigo::clock_at(C, (int)k) != t_last) return NAN;     // (the builder's own last value)
    return vigo::clock_at(C, (int)k);
}
double vigo_exact_pow_dd(double t, int d, int* ambiguous) {
    // the first tier as the sampler kernels run it: t^0 = 1, t^1 = t, then the running double-double product
    bool amb = false;
    double hi = d <= 0 ? 1.0 : t, lo = 0.0;
    for (int i = 2; i <= d && i <= 15; ++i) amb |= vigo::pow_step(hi, lo, t);
    if (ambiguous) *ambiguous = amb ? 1 : 0;
    return hi;
}
double vigo_exact_pow(double t, int d) {
    if (d < 0 || d > 15) return NAN;
    if (t == 0.0) return vigo::pow_exact(t, d);   // (the first tier does not track the sign of a zero result)
    int amb = 0;
    const double v = vigo_exact_pow_dd(t, d, &amb);
    return amb ? vigo::pow_exact(t, d) : v;
}
double vigo_exact_pow_integer(double t, int d) { return (d < 0 || d > 15) ? NAN : vigo::pow_exact(t, d); }
const char* vigo_build_arch(void) { return "gfx950"; }
const char* vigo_build_id(void) { return VIGO_BUILD_ID; }

void vigo_default_params(vigo_params_t* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    // cfg/bspline_interactive/bspline_planner_param.yaml:4-19, BT.h:46-47
    p->dthresh = 0.5;
    p->dist_thresh_dynamic = 0.5;
    p->ts_ctrl = 0.2;
    p->ts = 0.1;
    p->pred_horizon = 2.0;
    p->uncertain_factor = 1.0;
    p->w_distance = 1.0;
    p->w_smoothness = 1.0;
    p->w_feasibility = 1.0;
    p->w_dynamic = 1.0;
    p->min_height = 0.7;
    p->max_height = 1.3;
    p->plan_in_z = 0;
    // BT.cpp:695-699 over LB:942-954
    p->mem_size = 16;
    p->max_iterations = 200;
    p->max_linesearch = 40;
    p->past = 0;
    p->g_epsilon = 0.01;
    p->delta = 1e-5;
    p->min_step = 1e-20;
    p->max_step = 1e20;
    p->f_dec_coeff = 1e-4;
    p->s_curv_coeff = 0.9;
    p->xtol = 1.0e-16;
}

int vigo_create(vigo_handle_t* out, int device_ordinal) {
    if (!out) return VIGO_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return VIGO_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= count) return VIGO_ERR_INVALID_ARG;
    if (hipSetDevice(device_ordinal) != hipSuccess) return VIGO_ERR_NO_DEVICE;
    vigo_context* h = new vigo_context();
    h->device = device_ordinal;
    vigo_default_params(&h->params);
    h->dc = vigo::make_dev_const(h->params);
    bool ok = hipMalloc(reinterpret_cast<void**>(&h->dc_dev), sizeof(vigo::DevConst)) == hipSuccess &&
              hipMemcpy(h->dc_dev, &h->dc, sizeof(vigo::DevConst), hipMemcpyHostToDevice) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&h->dc_stage), vigo_context::kDcSlots * sizeof(vigo::DevConst)) == hipSuccess;
    for (int i = 0; ok && i < vigo_context::kDcSlots; ++i)
        ok = hipEventCreateWithFlags(&h->dc_event[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)vigo_destroy(h);
        return VIGO_ERR_HIP;
    }
    // launch geometry of this handle's device (vigo_solver.hip decides one or two waves per SIMD from it)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0)
        h->launch.simd_count = 4 * cus;
    *out = h;
    return VIGO_OK;
}

int vigo_destroy(vigo_handle_t h) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    if (h->grid_planes) (void)hipFree(h->grid_planes);
    if (h->esdf) (void)hipFree(h->esdf);
    if (h->fit_pinvT) (void)hipFree(h->fit_pinvT);
    if (h->times_dev) (void)hipFree(h->times_dev);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->rebound_idx) (void)hipFree(h->rebound_idx);
    if (h->dc_dev) (void)hipFree(h->dc_dev);
    for (int i = 0; i < vigo_context::kDcSlots; ++i)
        if (h->dc_event[i]) (void)hipEventDestroy(h->dc_event[i]);
    if (h->dc_stage) (void)hipHostFree(h->dc_stage);
    delete h;
    return VIGO_OK;
}

int vigo_set_stream(vigo_handle_t h, void* hip_stream) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (s != h->stream) {
        // the handle's device-side state (solver constants, cached tables) is updated in stream order on the bound
        // stream: work still queued on the old stream must not see updates issued on the new one
        VIGO_HIP(h, hipStreamSynchronize(h->stream));
        h->stream = s;
    }
    return VIGO_OK;
}

int vigo_set_params(vigo_handle_t h, const vigo_params_t* p) {
    if (!h || !p) return VIGO_ERR_INVALID_ARG;
    if (p->mem_size <= 0 || p->mem_size > VIGO_MAX_MEM_SIZE) return fail(h, VIGO_ERR_UNSUPPORTED, "mem_size outside [1, VIGO_MAX_MEM_SIZE]");
    if (p->past != 0) return fail(h, VIGO_ERR_UNSUPPORTED, "past != 0 (delta test) is not implemented; the reference runs past = 0");
    // lbfgs.hpp treats max_iterations == 0 as "until convergence or error" (LB:127-131); a device kernel
    // must have a bound every wave reaches, so the unbounded setting is refused (the reference runs 200)
    if (p->max_iterations == 0) return fail(h, VIGO_ERR_UNSUPPORTED, "max_iterations == 0 (unbounded) is not supported on the device");
    // (finite: an infinite knot spacing makes the span search of the spline evaluation spin on NaN knots)
    if (!(p->ts > 0) || !(p->ts_ctrl > 0) || !(p->ts < 1e9) || !(p->ts_ctrl < 1e9)) return fail(h, VIGO_ERR_INVALID_ARG, "ts and ts_ctrl must be finite and > 0");
    // every device loop needs a bound a wave reaches in reasonable time (the reference runs 200 / 40)
    if (p->max_iterations > 1000000 || p->max_linesearch > 100000)
        return fail(h, VIGO_ERR_UNSUPPORTED, "max_iterations > 1e6 or max_linesearch > 1e5");
    // predictionNum = int(predHorizon / ts) divides n in the dynamic-obstacle term (BT.cpp:1006, :1020): 0 is a
    // division by zero in the reference; a huge value is an unbounded device loop
    if (!(p->pred_horizon / p->ts >= 1.0) || !(p->pred_horizon / p->ts <= 100000.0))
        return fail(h, VIGO_ERR_INVALID_ARG, "pred_horizon / ts must lie in [1, 1e5]");
    // the argument checks of lbfgs_optimize (LB:1060-1104): refuse here rather than per trajectory
    if (p->g_epsilon < 0. || p->delta < 0. || p->min_step < 0. || p->max_step < p->min_step ||
        p->f_dec_coeff < 0. || p->s_curv_coeff <= p->f_dec_coeff || 1. <= p->s_curv_coeff ||
        p->xtol < 0. || p->max_linesearch <= 0 || p->max_iterations < 0)
        return fail(h, VIGO_ERR_INVALID_ARG, "invalid L-BFGS parameter (see lbfgs.hpp:1060-1104)");
    if (memcmp(&h->params, p, sizeof(*p)) == 0) return VIGO_OK;   // unchanged (callers re-send them per round): no copy, no sync
    // The kernels read the constants through dc_dev for the whole solve, and launches are asynchronous: the
    // refresh is therefore ordered WITH the bound stream (an async copy from a pinned staging slot), after every
    // solve already queued there and before every later one — a blocking copy on the null stream would change
    // max_iterations, the weights, ... under a solve still running on a non-blocking stream.
    const int slot = h->dc_next;
    VIGO_HIP(h, hipEventSynchronize(h->dc_event[slot]));          // the slot's previous copy (4 refreshes ago) has left it
    h->dc_stage[slot] = vigo::make_dev_const(*p);
    VIGO_HIP(h, hipMemcpyAsync(h->dc_dev, &h->dc_stage[slot], sizeof(vigo::DevConst), hipMemcpyHostToDevice, h->stream));
    VIGO_HIP(h, hipEventRecord(h->dc_event[slot], h->stream));
    h->dc_next = (slot + 1) % vigo_context::kDcSlots;
    h->params = *p;
    h->dc = h->dc_stage[slot];
    return VIGO_OK;
}

int vigo_get_params(vigo_handle_t h, vigo_params_t* p) {
    if (!h || !p) return VIGO_ERR_INVALID_ARG;
    *p = h->params;
    return VIGO_OK;
}

int vigo_set_precision(vigo_handle_t h, int precision) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (precision != VIGO_PREC_F64 && precision != VIGO_PREC_F32 && precision != VIGO_PREC_F64_FAST) return fail(h, VIGO_ERR_INVALID_ARG, "unknown precision");
    h->precision = precision;
    return VIGO_OK;
}

const char* vigo_last_error(vigo_handle_t h) { return h ? h->last_error.c_str() : "null handle"; }

/* ---- voxel map ---------------------------------------------------------------------- */

size_t vigo_grid_packed_bytes(int nx, int ny, int nz) {
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
    return (size_t)3 * nx * ny * ((nz + 31) / 32) * sizeof(uint32_t);
}

int vigo_pack_grid(vigo_handle_t h, int nx, int ny, int nz, const uint8_t* voxels_dev, uint32_t* packed_dev) {
    if (!h || !voxels_dev || !packed_dev || nx <= 0 || ny <= 0 || nz <= 0) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_pack_grid: bad argument");
    VIGO_HIP(h, (hipError_t)vigo::launch_pack_grid(h->stream, nx, ny, nz, voxels_dev, packed_dev));
    return VIGO_OK;
}

int vigo_inflate_grid(vigo_handle_t h, int nx, int ny, int nz, uint8_t* voxels_dev, int rx, int ry, int rz) {
    if (!h || !voxels_dev || nx <= 0 || ny <= 0 || nz <= 0 || rx < 0 || ry < 0 || rz < 0 || (size_t)nx * ny * nz > ((size_t)1 << 31))
        return fail(h, VIGO_ERR_INVALID_ARG, "vigo_inflate_grid: bad argument");
    if (rz > 31) return fail(h, VIGO_ERR_UNSUPPORTED, "vigo_inflate_grid: rz > 31 voxels");
    const size_t nw = (size_t)nx * ny * ((nz + 31) / 32);
    int rc = ensure_scratch(h, 2 * nw * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t* t = static_cast<uint32_t*>(h->scratch);
    VIGO_HIP(h, (hipError_t)vigo::launch_inflate(h->stream, nx, ny, nz, voxels_dev, t, t + nw, rx, ry, rz));
    return VIGO_OK;
}

int vigo_set_grid(vigo_handle_t h, int nx, int ny, int nz, const double origin[3], double res, const uint8_t* voxels_dev) {
    if (!h || !voxels_dev || !origin || nx <= 0 || ny <= 0 || nz <= 0 || !geometry_ok(origin, res)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_set_grid: bad argument");
    int rc = ensure_grid_storage(h, nx, ny, nz);
    if (rc) return rc;
    VIGO_HIP(h, (hipError_t)vigo::launch_pack_grid(h->stream, nx, ny, nz, voxels_dev, h->grid_planes));
    fill_grid_view(h, nx, ny, nz, origin, res);
    return VIGO_OK;
}

int vigo_set_grid_host(vigo_handle_t h, int nx, int ny, int nz, const double origin[3], double res, const uint8_t* voxels_host) {
    if (!h || !voxels_host || !origin || nx <= 0 || ny <= 0 || nz <= 0 || !geometry_ok(origin, res)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_set_grid_host: bad argument");
    size_t bytes = (size_t)nx * ny * nz;
    int rc = ensure_scratch(h, bytes);
    if (rc) return rc;
    VIGO_HIP(h, hipMemcpyAsync(h->scratch, voxels_host, bytes, hipMemcpyHostToDevice, h->stream));
    rc = vigo_set_grid(h, nx, ny, nz, origin, res, static_cast<const uint8_t*>(h->scratch));
    if (rc) return rc;
    VIGO_HIP(h, hipStreamSynchronize(h->stream));
    return VIGO_OK;
}

int vigo_set_grid_packed(vigo_handle_t h, int nx, int ny, int nz, const double origin[3], double res, const uint32_t* packed_dev) {
    if (!h || !packed_dev || !origin || nx <= 0 || ny <= 0 || nz <= 0 || !geometry_ok(origin, res)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_set_grid_packed: bad argument");
    int rc = ensure_grid_storage(h, nx, ny, nz);
    if (rc) return rc;
    VIGO_HIP(h, hipMemcpyAsync(h->grid_planes, packed_dev, vigo_grid_packed_bytes(nx, ny, nz), hipMemcpyDeviceToDevice, h->stream));
    fill_grid_view(h, nx, ny, nz, origin, res);
    return VIGO_OK;
}

int vigo_set_metric_bounds(vigo_handle_t h, const double bmin[3], const double bmax[3]) {
    if (!h || !bmin || !bmax) return VIGO_ERR_INVALID_ARG;
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_set_metric_bounds before vigo_set_grid");
    for (int a = 0; a < 3; ++a) { h->grid.bmin[a] = bmin[a]; h->grid.bmax[a] = bmax[a]; }
    return VIGO_OK;
}

int vigo_query_points(vigo_handle_t h, int which, int64_t Q, const double* pts, uint8_t* out) {
    if (!h || Q < 0 || (Q > 0 && (!pts || !out)) || which < 0 || which > 1) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_query_points: bad argument");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_query_points before vigo_set_grid");
    VIGO_HIP(h, (hipError_t)vigo::launch_query_points(h->stream, h->grid, which, Q, pts, 3, out));
    return VIGO_OK;
}

int vigo_guides_unknown(vigo_handle_t h, int64_t G, const double* guide_pv, uint8_t* out_unk) {
    if (!h || G < 0 || (G > 0 && (!guide_pv || !out_unk))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_guides_unknown: bad argument");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_guides_unknown before vigo_set_grid");
    VIGO_HIP(h, (hipError_t)vigo::launch_query_points(h->stream, h->grid, 1, G, guide_pv, 6, out_unk));
    return VIGO_OK;
}

int vigo_check_lists(vigo_handle_t h, int B, int N, const int32_t* guide_off, int64_t G, const int32_t* obs_off, int64_t O) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || N < 1 || G < 0 || O < 0) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_check_lists: bad argument");
    if (B == 0 || (!guide_off && !obs_off)) return 0;
    int rc = ensure_scratch(h, 64);
    if (rc) return rc;
    int* bad = static_cast<int*>(h->scratch);
    VIGO_HIP(h, (hipError_t)vigo::launch_check_lists(h->stream, B, N, guide_off, G, obs_off, O, bad));
    int host_bad = 0;
    VIGO_HIP(h, hipMemcpyAsync(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    VIGO_HIP(h, hipStreamSynchronize(h->stream));
    return host_bad;
}

/* ---- ViGO cost / gradient / solve ----------------------------------------------------- */

int vigo_cost_grad(vigo_handle_t h, int B, int N, const double* ctrl, const int32_t* guide_off,
                   const double* guide_pv, const uint8_t* guide_unk, const int32_t* obs_off,
                   const double* obs, int n_obs_shared, const double* weights, double* out_cost,
                   double* out_grad, double* out_terms) {
    int rc = check_solve_args(h, B, N, ctrl);
    if (rc) return rc;
    rc = check_list_args(h, guide_off, guide_pv, obs_off, obs, n_obs_shared);
    if (rc) return rc;
    SolveArgs a{};
    a.B = B; a.N = N;
    a.ctrl = const_cast<double*>(ctrl);
    a.guide_off = guide_off; a.guide_pv = guide_pv; a.guide_unk = guide_unk;
    a.obs_off = obs_off; a.obs = obs; a.n_obs_shared = n_obs_shared;
    a.weights = weights;
    a.out_cost = out_cost; a.out_grad = out_grad; a.out_terms = out_terms;
    VIGO_HIP(h, (hipError_t)vigo::launch_cost_grad(h->stream, a, h->dc, h->dc_dev, h->precision));
    return VIGO_OK;
}

int vigo_optimize(vigo_handle_t h, int B, int N, double* ctrl, const int32_t* guide_off,
                  const double* guide_pv, const uint8_t* guide_unk, const int32_t* obs_off,
                  const double* obs, int n_obs_shared, const double* weights, double* out_x,
                  int32_t* out_status, double* out_fx, int32_t* out_iters, int32_t* out_evals) {
    int rc = check_solve_args(h, B, N, ctrl);
    if (rc) return rc;
    rc = check_list_args(h, guide_off, guide_pv, obs_off, obs, n_obs_shared);
    if (rc) return rc;
    SolveArgs a{};
    a.B = B; a.N = N;
    a.ctrl = ctrl;
    a.guide_off = guide_off; a.guide_pv = guide_pv; a.guide_unk = guide_unk;
    a.obs_off = obs_off; a.obs = obs; a.n_obs_shared = n_obs_shared;
    a.weights = weights;
    a.out_x = out_x; a.out_status = out_status; a.out_fx = out_fx;
    a.out_iters = out_iters; a.out_evals = out_evals;
    if (vigo::optimize_lds_requirement(N, h->params.mem_size, h->precision) > (size_t)160 * 1024)
        return fail(h, VIGO_ERR_UNSUPPORTED_N, "the L-BFGS history of N control points x mem_size does not fit the 160 KiB LDS of a CU");
    VIGO_HIP(h, (hipError_t)vigo::launch_optimize(h->stream, a, h->dc, h->dc_dev, h->precision, h->launch));
    return VIGO_OK;
}

int vigo_rebound_rounds(vigo_handle_t h, int B, int N, double* ctrl, const int32_t* guide_off, const double* guide_pv,
                        const uint8_t* guide_unk, const int32_t* obs_off, const double* obs, int n_obs_shared, double* weights,
                        double gate_dt, double not_check_ratio, int max_rounds, vigo_rebound_state_t* state) {
    int rc = check_solve_args(h, B, N, ctrl);
    if (rc) return rc;
    rc = check_list_args(h, guide_off, guide_pv, obs_off, obs, n_obs_shared);
    if (rc) return rc;
    if (B > 0 && (!weights || !state)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_rebound_rounds: weights and state are required");
    if (max_rounds < 0 || max_rounds > 64 || !(not_check_ratio >= 0.0 && not_check_ratio < 1.0))
        return fail(h, VIGO_ERR_INVALID_ARG, "vigo_rebound_rounds: max_rounds outside [0, 64] or not_check_ratio outside [0, 1)");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_rebound_rounds before vigo_set_grid");
    if (vigo::optimize_lds_requirement(N, h->params.mem_size, h->precision) > (size_t)160 * 1024)
        return fail(h, VIGO_ERR_UNSUPPORTED_N, "the L-BFGS history of N control points x mem_size does not fit the 160 KiB LDS of a CU");
    if (B == 0) return VIGO_OK;
    int T = 0;
    const double* times = nullptr;
    rc = upload_sample_times(h, (N - 3) * h->params.ts_ctrl, gate_dt, &T, &times);
    if (rc) return rc;
    // compacted active set: B indices + the count (its own allocation: the scratch buffer serves other entry points)
    if ((size_t)B + 16 > h->rebound_cap) {
        if (h->rebound_idx) (void)hipFree(h->rebound_idx);
        h->rebound_idx = nullptr;
        h->rebound_cap = 0;
        VIGO_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->rebound_idx), ((size_t)B + 16) * sizeof(int32_t) + 64));
        h->rebound_cap = (size_t)B + 16;
    }
    int32_t* idx = h->rebound_idx + 16;
    int32_t* count = h->rebound_idx;          // flags[0]; flags[1], flags[2]: "a trajectory waits for the host"
    VIGO_HIP(h, hipMemsetAsync(count, 0, 16 * sizeof(int32_t), h->stream));
    SolveArgs a{};
    a.B = B; a.N = N;
    a.ctrl = ctrl;
    a.guide_off = guide_off; a.guide_pv = guide_pv; a.guide_unk = guide_unk;
    a.obs_off = obs_off; a.obs = obs; a.n_obs_shared = n_obs_shared;
    a.weights = weights;
    a.active_idx = idx; a.active_count = count;
    a.out_status = &state[0].lbfgs_status;
    a.status_stride = (int)(sizeof(vigo_rebound_state_t) / sizeof(int32_t));
    vigo::ReboundArgs r{};
    r.B = B; r.N = N; r.ctrl = ctrl;
    r.guide_off = guide_off; r.guide_pv = guide_pv;
    r.obs_off = obs_off; r.obs = obs; r.n_obs_shared = n_obs_shared;
    r.weights = weights; r.state = state;
    r.ts_ctrl = h->params.ts_ctrl; r.T = T; r.times = times;
    // the static gate stops at (1 - notCheckRatio_) * duration (BT.h:313); the dynamic one samples the whole
    // trajectory (evalTraj(), BT.h:345) — a prefix of the same clock
    r.T_static = count_sample_times(gate_dt, (1.0 - not_check_ratio) * ((N - 3) * h->params.ts_ctrl));
    if (r.T_static < 0 || r.T_static > T) r.T_static = T;
    r.dthresh = h->params.dthresh; r.not_check_ratio = not_check_ratio;
    r.flags = count;
    // the optimize() a trajectory still owes (BT.cpp:612 / after a host-side re-guide) ...
    VIGO_HIP(h, (hipError_t)vigo::launch_rebound_compact(h->stream, B, state, 0, idx, count));
    VIGO_HIP(h, (hipError_t)vigo::launch_optimize(h->stream, a, h->dc, h->dc_dev, h->precision, h->launch));
    // ... then the rounds: gates + decision, compaction of the still-active set, optimize
    for (int round = 0; round < max_rounds; ++round) {
        VIGO_HIP(h, (hipError_t)vigo::launch_rebound_decide(h->stream, h->grid, r));
        VIGO_HIP(h, (hipError_t)vigo::launch_rebound_compact(h->stream, B, state, 1, idx, count));
        VIGO_HIP(h, (hipError_t)vigo::launch_optimize(h->stream, a, h->dc, h->dc_dev, h->precision, h->launch));
    }
    return VIGO_OK;
}

/* ---- B-spline fit, evaluation and gates ------------------------------------------------- */

int vigo_bspline_fit(vigo_handle_t h, int B, int K, double ts, const double* points, const double* conds, double* ctrl_out) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || !(ts > 0) || (B > 0 && (!points || !ctrl_out))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_bspline_fit: bad argument");
    // bspline.cpp:83-87 needs at least 4 points; K + 2 control points must fit the solver's range
    if (K < 4 || K + 2 > VIGO_MAX_CTRL_POINTS) return fail(h, VIGO_ERR_UNSUPPORTED_N, "K outside [4, VIGO_MAX_CTRL_POINTS - 2]");
    if (h->fit_K != K || h->fit_ts != ts) {
        // one-off per (K, ts): factorise A on the device, keep the least-squares operator
        const size_t need = vigo::fit_pinv_doubles(K);
        if (need > h->fit_capacity) {
            if (h->fit_pinvT) (void)hipFree(h->fit_pinvT);
            h->fit_pinvT = nullptr;
            h->fit_capacity = 0;
            h->fit_K = 0;
            VIGO_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->fit_pinvT), need * sizeof(double)));
            h->fit_capacity = need;
        }
        int rc = ensure_scratch(h, vigo::fit_work_doubles(K) * sizeof(double));
        if (rc) return rc;
        h->fit_K = 0;
        VIGO_HIP(h, (hipError_t)vigo::launch_fit_setup(h->stream, K, ts, static_cast<double*>(h->scratch), h->fit_pinvT));
        h->fit_K = K;
        h->fit_ts = ts;
    }
    VIGO_HIP(h, (hipError_t)vigo::launch_bspline_fit(h->stream, B, K, h->fit_pinvT, points, conds, ctrl_out));
    return VIGO_OK;
}

int vigo_bspline_eval(vigo_handle_t h, int B, int N, const double* ctrl, int deriv, int T,
                      const double* times, double* out) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || T < 0 || deriv < 0 || deriv > 2 || ((B > 0 && T > 0) && (!ctrl || !times || !out)))
        return fail(h, VIGO_ERR_INVALID_ARG, "vigo_bspline_eval: bad argument");
    if (N < 4 || N > VIGO_MAX_CTRL_POINTS) return fail(h, VIGO_ERR_UNSUPPORTED_N, "N outside [4, VIGO_MAX_CTRL_POINTS]");
    VIGO_HIP(h, (hipError_t)vigo::launch_bspline_eval(h->stream, B, N, ctrl, h->params.ts_ctrl, deriv, T, times, out));
    return VIGO_OK;
}

int vigo_traj_collision(vigo_handle_t h, int B, int N, const double* ctrl, double dt,
                        uint8_t* out_flag, int32_t* out_first) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!ctrl || !out_flag))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_traj_collision: bad argument");
    if (N < 4 || N > VIGO_MAX_CTRL_POINTS) return fail(h, VIGO_ERR_UNSUPPORTED_N, "N outside [4, VIGO_MAX_CTRL_POINTS]");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_traj_collision before vigo_set_grid");
    int T = 0;
    const double* times = nullptr;
    double duration = (N - 3) * h->params.ts_ctrl;  // knots(N), BS.cpp:27
    int rc = upload_sample_times(h, (1.0 - 0.0) * duration, dt, &T, &times);
    if (rc) return rc;
    VIGO_HIP(h, (hipError_t)vigo::launch_traj_collision(h->stream, h->grid, B, N, ctrl, h->params.ts_ctrl, T, times, out_flag, out_first));
    return VIGO_OK;
}

int vigo_traj_dynamic_collision(vigo_handle_t h, int B, int N, const double* ctrl, double dt,
                                const int32_t* obs_off, const double* obs, int n_obs_shared,
                                uint8_t* out_flag) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!ctrl || !out_flag)) || n_obs_shared < 0) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_traj_dynamic_collision: bad argument");
    if (N < 4 || N > VIGO_MAX_CTRL_POINTS) return fail(h, VIGO_ERR_UNSUPPORTED_N, "N outside [4, VIGO_MAX_CTRL_POINTS]");
    int rc = check_list_args(h, nullptr, nullptr, obs_off, obs, n_obs_shared);
    if (rc) return rc;
    int T = 0;
    const double* times = nullptr;
    double duration = (N - 3) * h->params.ts_ctrl;
    rc = upload_sample_times(h, duration, dt, &T, &times);
    if (rc) return rc;
    VIGO_HIP(h, (hipError_t)vigo::launch_traj_dynamic_collision(h->stream, B, N, ctrl, h->params.ts_ctrl, T, times, obs_off, obs, n_obs_shared, out_flag));
    return VIGO_OK;
}

int vigo_ctrl_occupancy(vigo_handle_t h, int B, int N, const double* ctrl, uint8_t* out_pt, uint8_t* out_line) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (B < 0 || N < 1 || (B > 0 && (!ctrl || !out_pt || !out_line))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_ctrl_occupancy: bad argument");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_ctrl_occupancy before vigo_set_grid");
    VIGO_HIP(h, (hipError_t)vigo::launch_ctrl_occupancy(h->stream, h->grid, B, N, ctrl, out_pt, out_line));
    return VIGO_OK;
}

/* ---- min-snap QP and corridor checker ---------------------------------------------------- */

int vigo_minsnap(vigo_handle_t h, int T, int W, int deg, int diff, int cont, double desired_vel, double corridor_res,
                 const double* waypoints, const double* corridor, const double* conds, double* out_coeffs,
                 double* out_knots, int32_t* out_status) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (T < 0 || !(desired_vel > 0) || (corridor && !(corridor_res > 0)) || (T > 0 && (!waypoints || !out_coeffs || !out_knots || !out_status)))
        return fail(h, VIGO_ERR_INVALID_ARG, "vigo_minsnap: bad argument");
    if (deg != 7 || diff < 1 || diff > deg) return fail(h, VIGO_ERR_UNSUPPORTED, "vigo_minsnap: degree 7 polynomials only (polynomial_degree of cfg/planner*.yaml)");
    if (W < 2 || W > vigo::minsnap_max_waypoints()) return fail(h, VIGO_ERR_UNSUPPORTED_N, "vigo_minsnap: 2..11 waypoints per path on the device");
    const int K = W - 1;
    const int me = (2 + (K - 1) + (K - 1)) + 2 * (2 + (K - 1)) + (K - 1) * (cont - 2);
    if (cont < 2 || me > 64 || me > K * 8 || K * 8 - me > 40 || vigo::minsnap_lds_bytes(W, cont) > (size_t)160 * 1024)
        return fail(h, VIGO_ERR_UNSUPPORTED, "vigo_minsnap: continuity degree outside what one wavefront / 160 KiB of LDS holds");
    VIGO_HIP(h, (hipError_t)vigo::launch_minsnap(h->stream, T, W, deg, diff, cont, desired_vel, corridor_res, waypoints, corridor, conds,
                                                 out_coeffs, out_knots, out_status, h->launch));
    return VIGO_OK;
}


int vigo_corridor_check(vigo_handle_t h, int S, int deg, const double* coeffs, const int32_t* n_samp,
                        const double* delT, const double box[3], double map_res, uint8_t* out_flag,
                        int32_t* out_first, int32_t* out_count) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (S < 0 || deg < 0 || deg > 15 || !box || !(map_res > 0) || (S > 0 && (!coeffs || !n_samp || !delT || !out_flag)))
        return fail(h, VIGO_ERR_INVALID_ARG, "vigo_corridor_check: bad argument");
    if (!sweep_box_ok(box, map_res)) return fail(h, VIGO_ERR_UNSUPPORTED, "vigo_corridor_check: collision box not finite or more than 32768 lattice points per pose");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_corridor_check before vigo_set_grid");
    for (int a = 0; a < 3; ++a) {
        double q = h->grid.origin[a] / h->grid.res;
        if (fabs(q - floor(q + 0.5)) > 1e-6)
            return fail(h, VIGO_ERR_UNSUPPORTED, "corridor checker needs a grid origin that is a multiple of res (octomap keys)");
    }
    if (S == 0) return VIGO_OK;
    // scratch: the first pass' work list for the second, then (up to 16384 segments: 42 MB) the segments' clock tables
    const size_t todo_bytes = ((size_t)S * sizeof(int) + 255) & ~(size_t)255;
    const size_t clock_bytes = S <= 16384 ? vigo::corridor_clock_ws_bytes(S) : 0;
    int rc = ensure_scratch(h, todo_bytes + clock_bytes);
    if (rc != VIGO_OK) return rc;
    char* ws = static_cast<char*>(h->scratch);
    VIGO_HIP(h, (hipError_t)vigo::launch_corridor_check2(h->stream, h->grid, S, deg, coeffs, n_samp, delT, box, map_res, out_flag,
                                                         out_first, out_count, reinterpret_cast<int*>(ws),
                                                         clock_bytes ? ws + todo_bytes : nullptr));
    return VIGO_OK;
}

int vigo_poly_sample(vigo_handle_t h, int S, int deg, const double* coeffs, const int32_t* n_samp, const double* delT,
                     int stride, double* out_pos, float* out_pos_f32) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (S < 0 || deg < 0 || deg > 15 || stride < 0 || (S > 0 && stride > 0 && (!coeffs || !n_samp || !delT || (!out_pos && !out_pos_f32))))
        return fail(h, VIGO_ERR_INVALID_ARG, "vigo_poly_sample: bad argument");
    VIGO_HIP(h, (hipError_t)vigo::launch_poly_sample(h->stream, S, deg, coeffs, n_samp, delT, stride, out_pos, out_pos_f32));
    return VIGO_OK;
}

int vigo_box_collision_points(vigo_handle_t h, int64_t M, const double* pts, const double box[3], double map_res, uint8_t* out) {
    if (!h) return VIGO_ERR_INVALID_ARG;
    if (M < 0 || !box || !(map_res > 0) || (M > 0 && (!pts || !out))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_box_collision_points: bad argument");
    if (!sweep_box_ok(box, map_res)) return fail(h, VIGO_ERR_UNSUPPORTED, "vigo_box_collision_points: collision box not finite or more than 32768 lattice points per pose");
    if (!h->has_grid) return fail(h, VIGO_ERR_NO_GRID, "vigo_box_collision_points before vigo_set_grid");
    for (int a = 0; a < 3; ++a) {
        double q = h->grid.origin[a] / h->grid.res;
        if (fabs(q - floor(q + 0.5)) > 1e-6)
            return fail(h, VIGO_ERR_UNSUPPORTED, "corridor checker needs a grid origin that is a multiple of res (octomap keys)");
    }
    VIGO_HIP(h, (hipError_t)vigo::launch_box_points(h->stream, h->grid, M, pts, box, map_res, out));
    return VIGO_OK;
}

/* ---- ESDF ----------------------------------------------------------------------------------- */

int vigo_set_esdf(vigo_handle_t h, int nx, int ny, int nz, const double origin[3], double res, const float* dist_dev) {
    if (!h || !dist_dev || !origin || nx < 2 || ny < 2 || nz < 2 || !geometry_ok(origin, res)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_set_esdf: bad argument");
    if ((long long)nx * ny * nz > (1LL << 33)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_set_esdf: lattice too large");
    size_t bytes = vigo::esdf_bricked_floats(nx, ny, nz) * sizeof(float);   // 3.56x the lattice (one line per cell group)
    if (bytes > h->esdf_capacity) {
        if (h->esdf) (void)hipFree(h->esdf);
        h->esdf = nullptr;
        h->esdf_capacity = 0;
        VIGO_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->esdf), bytes));
        h->esdf_capacity = bytes;
    }
    VIGO_HIP(h, (hipError_t)vigo::launch_esdf_brick(h->stream, nx, ny, nz, dist_dev, h->esdf));   // row-major -> one line per cell group
    h->esdf_view.dist = h->esdf;
    h->esdf_view.nx = nx; h->esdf_view.ny = ny; h->esdf_view.nz = nz;
    h->esdf_view.nby = vigo::esdf_bricks_along(ny); h->esdf_view.nbz = vigo::esdf_bricks_along(nz);
    h->esdf_view.res = res;
    for (int a = 0; a < 3; ++a) h->esdf_view.origin[a] = origin[a];
    h->has_esdf = true;
    return VIGO_OK;
}

int vigo_esdf_query(vigo_handle_t h, int64_t Q, const double* pts, double* out_dist, double* out_grad) {
    if (!h || Q < 0 || (Q > 0 && (!pts || !out_dist || !out_grad))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_esdf_query: bad argument");
    if (!h->has_esdf) return fail(h, VIGO_ERR_NO_GRID, "vigo_esdf_query before vigo_set_esdf");
    VIGO_HIP(h, (hipError_t)vigo::launch_esdf_query(h->stream, h->esdf_view, Q, pts, out_dist, out_grad));
    return VIGO_OK;
}

int vigo_esdf_query_f32(vigo_handle_t h, int64_t Q, const float* pts, float* out_dist_grad) {
    if (!h || Q < 0 || (Q > 0 && (!pts || !out_dist_grad))) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_esdf_query_f32: bad argument");
    if (Q > 0 && (reinterpret_cast<uintptr_t>(out_dist_grad) & 15u)) return fail(h, VIGO_ERR_INVALID_ARG, "vigo_esdf_query_f32: out must be 16-byte aligned");
    if (!h->has_esdf) return fail(h, VIGO_ERR_NO_GRID, "vigo_esdf_query_f32 before vigo_set_esdf");
    VIGO_HIP(h, (hipError_t)vigo::launch_esdf_query_f32(h->stream, h->esdf_view, Q, pts, out_dist_grad));
    return VIGO_OK;
}

}  // extern "C"

// vigo_map.hip — voxel snapshot packing, point queries, B-spline evaluation and the
// rebound-loop gates (hasCollisionTrajectory, hasDynamicCollisionTrajectory, the map queries
// of findCollisionSeg).  All HBM-bound byte/bit work: coalesced reads, one bit per voxel per
// plane so a 256^3 plane is 2 MiB (L2-resident per XCD) and a 512^3 plane 16 MiB (MALL).
//
// These gates look each sample up directly in the packed planes (L2 hits): a trajectory makes
// ~120-240 lookups scattered over a 7 m path, fewer words than staging its bounding volume in
// LDS would read.  The LDS-tiled path is the corridor checker (vigo_corridor.hip), where one
// segment makes ~10^5 lookups inside a small volume.
#include "vigo_exact_time.hpp"
#include "vigo_grid.hpp"

namespace vigo {
namespace {

// one thread per output word triple: reads 32 voxels along z, writes one word per plane
__global__ void k_pack_grid(int nx, int ny, int nz, int nzw, const uint8_t* __restrict__ vox,
                            uint32_t* __restrict__ packed) {
    const size_t plane_words = (size_t)nx * ny * nzw;
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= plane_words) return;
    const size_t col = wi / nzw;
    const int w = (int)(wi % nzw);
    const uint8_t* src = vox + col * nz + (size_t)w * 32;
    const int cnt = min(32, nz - w * 32);
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    if (cnt == 32 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4 q[2] = {s4[0], s4[1]};
        const uint32_t* wds = reinterpret_cast<const uint32_t*>(q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t v = (wds[i] >> (8 * j)) & 0xFFu;
                const int bit = 4 * i + j;
                b0 |= (v & 1u) << bit;
                b1 |= ((v >> 1) & 1u) << bit;
                b2 |= ((v >> 2) & 1u) << bit;
            }
        }
    } else {
        for (int i = 0; i < cnt; ++i) {
            const uint32_t v = src[i];
            b0 |= (v & 1u) << i;
            b1 |= ((v >> 1) & 1u) << i;
            b2 |= ((v >> 2) & 1u) << i;
        }
    }
    packed[wi] = b0;
    packed[plane_words + wi] = b1;
    packed[2 * plane_words + wi] = b2;
}

// ---- inflation on bit planes -----------------------------------------------------------------
// bit0 of the byte grid := box dilation of bit2 (map_manager inflates the occupied voxels by the robot
// size).  The occupied bit is packed to one bit per voxel (z fastest, 32 voxels per
// word — the snapshot's plane format), dilated there — along z with shifts and carries between
// neighbouring words, along y and x by OR-ing whole words — and merged back: 3 bytes of HBM traffic
// per voxel instead of 2 per voxel and pass (a 256^3 plane is 2 MiB: the three dilation passes never
// leave the L2).
__global__ void k_pack_bit(int nz, int nzw, size_t n_words, int bit, const uint8_t* __restrict__ vox, uint32_t* __restrict__ plane) {
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const size_t col = wi / nzw;
    const int w = (int)(wi % nzw);
    const uint8_t* src = vox + col * nz + (size_t)w * 32;
    const int cnt = min(32, nz - w * 32);
    uint32_t b = 0;
    if (cnt == 32 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {   // 32 voxels = two 16-byte loads
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        const uint4 q[2] = {s4[0], s4[1]};
        const uint32_t* wds = reinterpret_cast<const uint32_t*>(q);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) b |= ((wds[i] >> (8 * j + bit)) & 1u) << (4 * i + j);
    } else {
        for (int i = 0; i < cnt; ++i) b |= (uint32_t)((src[i] >> bit) & 1u) << i;
    }
    plane[wi] = b;
}
// z: out word = OR over |d| <= r of the row shifted by d bits (r <= 31: one neighbour word each side)
__global__ void k_dilate_bits_z(int nz, int nzw, size_t n_words, int r, const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const int w = (int)(wi % nzw);
    const uint64_t lo = w > 0 ? in[wi - 1] : 0u, mid = in[wi], hi = w + 1 < nzw ? in[wi + 1] : 0u;
    uint32_t acc = (uint32_t)mid;
    for (int d = 1; d <= r; ++d) {
        acc |= (uint32_t)(((mid << 32 | lo) >> (32 - d)) & 0xffffffffu);   // voxels z - d land on z
        acc |= (uint32_t)(((hi << 32 | mid) >> d) & 0xffffffffu);          // voxels z + d land on z
    }
    const int valid = nz - 32 * w;                                          // bits past nz stay clear
    if (valid < 32) acc &= (1u << valid) - 1u;
    out[wi] = acc;
}
// y (stride nzw words, length ny) or x (stride ny * nzw, length nx): OR of whole words
__global__ void k_dilate_bits_rows(size_t n_words, size_t stride, int len, int r, const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const int c = (int)((wi / stride) % (size_t)len);
    const int lo = c - r < 0 ? -c : -r, hi = c + r >= len ? len - 1 - c : r;
    uint32_t acc = 0;
    for (int d = lo; d <= hi; ++d) acc |= in[(ptrdiff_t)wi + (ptrdiff_t)d * (ptrdiff_t)stride];
    out[wi] = acc;
}
// one plane word (32 voxels) per thread: bit0 of the 32 bytes from the word's bits
__global__ void k_merge_bit0(int nz, int nzw, size_t n_words, const uint32_t* __restrict__ plane, uint8_t* __restrict__ vox) {
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const size_t col = wi / nzw;
    const int w = (int)(wi % nzw);
    uint8_t* dst = vox + col * nz + (size_t)w * 32;
    const int cnt = min(32, nz - w * 32);
    const uint32_t bits = plane[wi];
    if (cnt == 32 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        uint4 q[2] = {d4[0], d4[1]};
        uint32_t* wds = reinterpret_cast<uint32_t*>(q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t nib = (bits >> (4 * i)) & 0xFu;
            const uint32_t spread = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);   // one bit per byte
            wds[i] = (wds[i] & 0xFEFEFEFEu) | spread;
        }
        d4[0] = q[0];
        d4[1] = q[1];
    } else {
        for (int i = 0; i < cnt; ++i) dst[i] = (uint8_t)((dst[i] & ~1u) | ((bits >> i) & 1u));
    }
}

// CSR offsets off[n]: off[0] == 0, non-decreasing, off[n-1] <= total; every violation is counted
__global__ void k_check_offsets(const int32_t* __restrict__ off, int64_t n, int64_t total, int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = off[i];
    bool wrong = v < 0 || (int64_t)v > total;
    if (i == 0) wrong = wrong || v != 0;
    else wrong = wrong || v < off[i - 1];
    if (wrong) atomicAdd(bad, 1);
}

__global__ void k_query_points(GridView g, int plane, int64_t Q, const double* __restrict__ pts,
                               int stride, uint8_t* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const double* p = pts + q * stride;
    out[q] = (uint8_t)grid_plane_pos(g, plane, p[0], p[1], p[2]);
}

// grid: (ceil(T/256), B)
__global__ void k_bspline_eval(int B, int N, const double* __restrict__ ctrl, double ts, int deriv,
                               int T, const double* __restrict__ times, double* __restrict__ out) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T) return;
    double p[3];
    traj_eval(ctrl + (size_t)b * N * 3, N, ts, deriv, times[k], p);
    double* dst = out + ((size_t)b * T + k) * 3;
    dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
}

// one 64-lane wave per trajectory; lanes stride over the samples; first hit = wave min
// t_k of the reference's sample loop `for (t = 0; t <= tmax; t += dt)` (BT.h:313, :347), k = 0 .. T-1, by the
// closed form of vigo_exact_time.hpp — filled once per (dt, tmax) and cached in the handle
__global__ void k_fill_sample_times(double dt, int T, double* __restrict__ times) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < T) times[k] = accumulated_time(dt, k);
}

__global__ void __launch_bounds__(64) k_traj_collision(GridView g, int B, int N, const double* __restrict__ ctrl,
                                                       double ts, int T, const double* __restrict__ times,
                                                       uint8_t* __restrict__ out_flag, int32_t* __restrict__ out_first) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const double* c = ctrl + (size_t)b * N * 3;
    int first = 0x7fffffff;
    for (int k = threadIdx.x; k < T; k += 64) {
        double p[3];
        traj_eval(c, N, ts, 0, times[k], p);
        if (grid_plane_pos(g, 0, p[0], p[1], p[2])) { first = k; break; }  // k ascends per lane
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) first = min(first, __shfl_xor(first, m, 64));
    if (threadIdx.x == 0) {
        out_flag[b] = first != 0x7fffffff;
        if (out_first) out_first[b] = first != 0x7fffffff ? first : -1;
    }
}

__global__ void __launch_bounds__(64) k_traj_dynamic_collision(int B, int N, const double* __restrict__ ctrl, double ts,
                                                               int T, const double* __restrict__ times,
                                                               const int32_t* __restrict__ obs_off,
                                                               const double* __restrict__ obs, int n_obs_shared,
                                                               uint8_t* __restrict__ out_flag) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const double* c = ctrl + (size_t)b * N * 3;
    const int o0 = (obs && obs_off) ? obs_off[b] : 0;
    const int o1 = !obs ? 0 : (obs_off ? obs_off[b + 1] : n_obs_shared);
    int hit = 0;
    for (int k = threadIdx.x; k < T && !hit; k += 64) {
        double p[3];
        traj_eval(c, N, ts, 0, times[k], p);
        for (int j = o0; j < o1; ++j) {
            const double* o = obs + 9 * (size_t)j;
            const double size = fmin(o[6] / 2, o[7] / 2);  // BT.h:358 (min, unlike the cost term)
            const double dx = p[0] - o[0], dy = p[1] - o[1];
            const double dist = sqrt((dx * dx + dy * dy) + 0.0) - size;
            if (dist < 0) { hit = 1; break; }
        }
    }
    hit = __any(hit);
    if (threadIdx.x == 0) out_flag[b] = (uint8_t)(hit != 0);
}

// isInflatedOccupiedLine(q, p) of the dense map contract (include/vigo.h), given the point flags of both ends:
// end points, then int(dist / res) - 1 interior steps of length res from q
__device__ __forceinline__ unsigned line_occupied(const GridView& g, const double* q, const double* p, unsigned occ_q, unsigned occ_p) {
    unsigned line = occ_q | occ_p;
    if (!line) {
        const double d0 = p[0] - q[0], d1 = p[1] - q[1], d2 = p[2] - q[2];
        const double dist = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        const double i0 = d0 / dist * g.res, i1 = d1 / dist * g.res, i2 = d2 / dist * g.res;
        const int steps = (int)(dist / g.res);
        for (int s = 1; s < steps; ++s) {
            if (grid_plane_pos(g, 0, q[0] + s * i0, q[1] + s * i1, q[2] + s * i2)) { line = 1; break; }
        }
    }
    return line;
}

// one thread per control point: point flag and the line (i-1, i) flag
__global__ void k_ctrl_occupancy(GridView g, int B, int N, const double* __restrict__ ctrl,
                                 uint8_t* __restrict__ out_pt, uint8_t* __restrict__ out_line) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * N) return;
    const int i = (int)(idx % N);
    const double* p = ctrl + idx * 3;
    const unsigned occ = grid_plane_pos(g, 0, p[0], p[1], p[2]);
    out_pt[idx] = (uint8_t)occ;
    unsigned line = 0;
    if (i > 0) {
        const double* q = p - 3;  // previous control point
        line = line_occupied(g, q, p, grid_plane_pos(g, 0, q[0], q[1], q[2]), occ);
    }
    out_line[idx] = (uint8_t)line;
}

// ---- vigo_rebound_rounds: gates + the loop body of bsplineTraj::optimizeTrajectory (BT.cpp:619-679) ----------
// One 64-lane wave per ACTIVE trajectory.  The gates are the kernels above (lanes stride over the samples); the
// map queries of findCollisionSeg run one control point per lane; the segment bookkeeping, the comparison with the
// previous segments and the guide test are a serial scan by lane 0 over a few dozen bytes in LDS.
__global__ void __launch_bounds__(64) k_rebound_decide(GridView g, ReboundArgs A) {
    __shared__ uint8_t s_pt[VIGO_MAX_CTRL_POINTS], s_ln[VIGO_MAX_CTRL_POINTS];
    __shared__ int s_need_host;
    const int b = blockIdx.x;
    if (b >= A.B) return;
    // a trajectory of an EARLIER round waits for the host (A*): the queued rounds that follow are no-ops for the whole
    // batch, like the lock-step of the host-driven loop — the waiting trajectories are on the batch's critical path
    if (A.flags[1] != 0) return;
    vigo_rebound_state_t& st = A.state[b];
    if (st.status != VIGO_RB_ACTIVE) return;
    const int N = A.N, lane = threadIdx.x;
    const double* c = A.ctrl + (size_t)b * N * 3;

    // hasCollisionTrajectory, BT.h:307-325 (same walk as k_traj_collision)
    int col = 0;
    for (int k = lane; k < A.T_static && !col; k += 64) {
        double p[3];
        traj_eval(c, N, A.ts_ctrl, 0, A.times[k], p);
        if (grid_plane_pos(g, 0, p[0], p[1], p[2])) col = 1;
    }
    col = __any(col) ? 1 : 0;
    // hasDynamicCollisionTrajectory, BT.h:344-368 — only for a trajectory that has obstacles (BT.cpp:621-626)
    const int o0 = (A.obs && A.obs_off) ? A.obs_off[b] : 0;
    const int o1 = !A.obs ? 0 : (A.obs_off ? A.obs_off[b + 1] : A.n_obs_shared);
    int dyn = 0;
    if (o1 > o0) {
        for (int k = lane; k < A.T && !dyn; k += 64) {
            double p[3];
            traj_eval(c, N, A.ts_ctrl, 0, A.times[k], p);
            for (int j = o0; j < o1; ++j) {
                const double* o = A.obs + 9 * (size_t)j;
                const double size = fmin(o[6] / 2, o[7] / 2);
                const double dx = p[0] - o[0], dy = p[1] - o[1];
                if (sqrt((dx * dx + dy * dy) + 0.0) - size < 0) { dyn = 1; break; }
            }
        }
        dyn = __any(dyn) ? 1 : 0;
    }
    if (lane == 0) {
        st.gate_static = col;
        st.gate_dynamic = dyn;
        st.rounds += 1;
    }
    if (!col && !dyn) {                                   // BT.cpp:628-631
        if (lane == 0) st.status = VIGO_RB_DONE;
        return;
    }
    if (st.fail_count >= 4) {                             // BT.cpp:640-654: forced A* re-guide, the host's
        if (lane == 0) { st.status = VIGO_RB_NEEDS_HOST; A.flags[2] = 1; }
        return;
    }
    if (col) {
        // the map queries of findCollisionSeg (BT.cpp:412, :435), one control point per lane
        for (int i = lane; i < N; i += 64) {
            const double* p = c + 3 * i;
            s_pt[i] = (uint8_t)grid_plane_pos(g, 0, p[0], p[1], p[2]);
        }
        __syncthreads();
        for (int i = lane; i < N; i += 64)
            s_ln[i] = i > 0 ? (uint8_t)line_occupied(g, c + 3 * (i - 1), c + 3 * i, s_pt[i - 1], s_pt[i]) : 0;
        __syncthreads();
        if (lane == 0) {
            // findCollisionSeg, BT.cpp:403-445 (incl. the corner case that can duplicate a segment)
            int seg[2 * VIGO_MAX_COLLISION_SEGS];
            int n_new = 0;
            auto push = [&](int a, int e) {
                if (n_new < VIGO_MAX_COLLISION_SEGS) { seg[2 * n_new] = a; seg[2 * n_new + 1] = e; }
                ++n_new;
            };
            bool prev_has = false;
            const int end_idx = (int)((N - 3 - 1) - A.not_check_ratio * (N - 2 * 3));
            int pair_start = 3, pair_end = 3;
            for (int i = 3; i <= end_idx; ++i) {
                const bool has = s_pt[i] != 0;
                if (has != prev_has) {
                    if (has) pair_start = i - 1;
                    else { pair_end = i; push(pair_start, pair_end); }
                }
                if (has && i == end_idx - 1) { pair_end = N - 1; push(pair_start, pair_end); }
                if (i != 3 && !prev_has && !has && s_ln[i]) push(i - 1, i);
                prev_has = has;
            }
            bool need_host = n_new > VIGO_MAX_COLLISION_SEGS;
            // isReguideRequired, BT.cpp:573-608: a control point inside a new segment that no previous segment covers,
            // or a covered one that fails isControlPointRequireNewGuide (BT.h:417-429), asks for A*
            if (!need_host) {
                const int n_prev = min(max(st.n_seg, 0), (int)VIGO_MAX_COLLISION_SEGS);   // (device data: never index beyond the array)
                auto in_prev = [&](int i) {
                    for (int k = 0; k < n_prev; ++k)
                        if (i >= st.seg[2 * k] && i <= st.seg[2 * k + 1]) return true;
                    return false;
                };
                auto require_new_guide = [&](int i) {
                    if (!A.guide_off || !A.guide_pv) return true;
                    const double* ci = c + 3 * i;
                    for (int j = A.guide_off[(size_t)b * N + i]; j < A.guide_off[(size_t)b * N + i + 1]; ++j) {
                        const double* pv = A.guide_pv + 6 * (size_t)j;
                        const double dist = ((ci[0] - pv[0]) * pv[3] + (ci[1] - pv[1]) * pv[4]) + (ci[2] - pv[2]) * pv[5];
                        if (A.dthresh - dist > 0) return false;
                    }
                    return true;
                };
                auto asks = [&](int i) { return !in_prev(i) || require_new_guide(i); };
                for (int k = 0; k < n_new && !need_host; ++k) {
                    const int a = seg[2 * k], e = seg[2 * k + 1];
                    for (int i = a + 1; i <= e - 1 && !need_host; ++i) need_host = asks(i);
                    if (e - a - 1 == 0)
                        for (int i = a; i <= e && !need_host; ++i) need_host = asks(i);
                }
            }
            s_need_host = need_host ? 1 : 0;
            if (need_host) {
                st.status = VIGO_RB_NEEDS_HOST;           // untouched state: the host repeats the step with its own A*
                A.flags[2] = 1;
            } else {
                st.n_seg = n_new;                         // collisionSeg_ = the new segments (BT.cpp:575)
                for (int k = 0; k < 2 * n_new; ++k) st.seg[k] = seg[k];
                A.weights[4 * (size_t)b + 0] *= 2.0;      // BT.cpp:672
                st.fail_count += 1;
            }
        }
        __syncthreads();
        if (s_need_host) return;
    }
    if (dyn && lane == 0) A.weights[4 * (size_t)b + 3] *= 2.0;   // BT.cpp:677-679
}

// ascending indices of the trajectories a launch works on: one block, a scan over per-thread counts
// flags: [0] = count (out), [1] = "a trajectory waits for the host since an earlier round" (read by this round's
// kernels), [2] = the same as raised by this round's decide pass; mode 1 publishes [2] into [1] for the next round
__global__ void __launch_bounds__(1024) k_rebound_compact(int B, vigo_rebound_state_t* __restrict__ state, int mode,
                                                          int32_t* __restrict__ idx, int32_t* __restrict__ flags) {
    __shared__ int s_cnt[1024];
    const int tid = threadIdx.x;
    int32_t* count = flags;
    if (mode == 1 && flags[1] != 0) {            // (uniform: every thread reads the same word)
        if (tid == 0) *count = 0;
        return;
    }
    if (mode == 1 && flags[2] != 0) {
        // this round's decisions hand a trajectory to the host: the optimize() the still-active trajectories owe is
        // deferred to the caller's next call (solve_first), where it shares ONE launch with the re-guided ones —
        // two launches in a row would put two solve latencies on the batch's critical path
        const int per_ = (B + 1023) / 1024;
        for (int b = min(B, tid * per_); b < min(B, tid * per_ + per_); ++b)
            if (state[b].status == VIGO_RB_ACTIVE) state[b].solve_first = 1;
        if (tid == 0) { *count = 0; flags[1] = 1; }
        return;
    }
    const int per = (B + 1023) / 1024;
    const int lo = min(B, tid * per), hi = min(B, lo + per);
    auto wanted = [&](int b) {
        return state[b].status == VIGO_RB_ACTIVE && (mode == 1 || state[b].solve_first != 0);
    };
    int n = 0;
    for (int b = lo; b < hi; ++b) n += wanted(b) ? 1 : 0;
    s_cnt[tid] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {            // inclusive scan
        const int v = tid >= off ? s_cnt[tid - off] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    int at = s_cnt[tid] - n;
    for (int b = lo; b < hi; ++b) {
        if (wanted(b)) {
            idx[at++] = b;
            if (mode == 0) state[b].solve_first = 0;
        }
    }
    if (tid == 1023) {
        *count = s_cnt[1023];
        if (mode == 1) flags[1] = flags[2];
    }
}

}  // namespace

int launch_pack_grid(hipStream_t s, int nx, int ny, int nz, const uint8_t* vox, uint32_t* packed) {
    const int nzw = (nz + 31) / 32;
    const size_t words = (size_t)nx * ny * nzw;
    const int block = 256;
    const size_t grid = (words + block - 1) / block;
    hipLaunchKernelGGL(k_pack_grid, dim3((unsigned)grid), dim3(block), 0, s, nx, ny, nz, nzw, vox, packed);
    return (int)hipGetLastError();
}

int launch_inflate(hipStream_t s, int nx, int ny, int nz, uint8_t* vox, uint32_t* planeA, uint32_t* planeB, int rx, int ry, int rz) {
    const int nzw = (nz + 31) / 32;
    const size_t nw = (size_t)nx * ny * nzw;
    const unsigned gw = (unsigned)((nw + 255) / 256);
    hipLaunchKernelGGL(k_pack_bit, dim3(gw), dim3(256), 0, s, nz, nzw, nw, 2, vox, planeA);
    hipLaunchKernelGGL(k_dilate_bits_z, dim3(gw), dim3(256), 0, s, nz, nzw, nw, rz, planeA, planeB);
    hipLaunchKernelGGL(k_dilate_bits_rows, dim3(gw), dim3(256), 0, s, nw, (size_t)nzw, ny, ry, planeB, planeA);
    hipLaunchKernelGGL(k_dilate_bits_rows, dim3(gw), dim3(256), 0, s, nw, (size_t)ny * nzw, nx, rx, planeA, planeB);
    hipLaunchKernelGGL(k_merge_bit0, dim3(gw), dim3(256), 0, s, nz, nzw, nw, planeB, vox);
    return (int)hipGetLastError();
}

int launch_check_lists(hipStream_t s, int B, int N, const int32_t* guide_off, int64_t G, const int32_t* obs_off, int64_t O,
                       int* bad) {
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    const int block = 256;
    if (guide_off) {
        const int64_t n = (int64_t)B * N + 1;
        hipLaunchKernelGGL(k_check_offsets, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, s, guide_off, n, G, bad);
    }
    if (obs_off) {
        const int64_t n = (int64_t)B + 1;
        hipLaunchKernelGGL(k_check_offsets, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, s, obs_off, n, O, bad);
    }
    return (int)hipGetLastError();
}

int launch_query_points(hipStream_t s, const GridView& g, int which, int64_t Q, const double* pts,
                        int pt_stride, uint8_t* out) {
    if (Q <= 0) return hipSuccess;
    const int block = 256;
    hipLaunchKernelGGL(k_query_points, dim3((unsigned)((Q + block - 1) / block)), dim3(block), 0, s, g, which, Q, pts,
                       pt_stride, out);
    return (int)hipGetLastError();
}

int launch_bspline_eval(hipStream_t s, int B, int N, const double* ctrl, double ts_ctrl, int deriv,
                        int T, const double* times, double* out) {
    if (B <= 0 || T <= 0) return hipSuccess;
    const int block = 64;
    // gridDim.y is limited to 65535: larger batches go out in slices of that many trajectories
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = (B - b0) < 65535 ? (B - b0) : 65535;
        hipLaunchKernelGGL(k_bspline_eval, dim3((T + block - 1) / block, nb), dim3(block), 0, s, nb, N, ctrl + (size_t)b0 * N * 3,
                           ts_ctrl, deriv, T, times, out + (size_t)b0 * T * 3);
    }
    return (int)hipGetLastError();
}

int launch_traj_collision(hipStream_t s, const GridView& g, int B, int N, const double* ctrl,
                          double ts_ctrl, int T, const double* times, uint8_t* out_flag,
                          int32_t* out_first) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_traj_collision, dim3(B), dim3(64), 0, s, g, B, N, ctrl, ts_ctrl, T, times, out_flag, out_first);
    return (int)hipGetLastError();
}

int launch_traj_dynamic_collision(hipStream_t s, int B, int N, const double* ctrl, double ts_ctrl,
                                  int T, const double* times, const int32_t* obs_off,
                                  const double* obs, int n_obs_shared, uint8_t* out_flag) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_traj_dynamic_collision, dim3(B), dim3(64), 0, s, B, N, ctrl, ts_ctrl, T, times, obs_off, obs,
                       n_obs_shared, out_flag);
    return (int)hipGetLastError();
}

int launch_fill_sample_times(hipStream_t s, double dt, int T, double* times) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_sample_times, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, dt, T, times);
    return (int)hipGetLastError();
}

int launch_rebound_decide(hipStream_t s, const GridView& g, const ReboundArgs& a) {
    if (a.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rebound_decide, dim3(a.B), dim3(64), 0, s, g, a);
    return (int)hipGetLastError();
}

int launch_rebound_compact(hipStream_t s, int B, vigo_rebound_state_t* state, int mode, int32_t* idx, int32_t* flags) {
    hipLaunchKernelGGL(k_rebound_compact, dim3(1), dim3(1024), 0, s, B, state, mode, idx, flags);
    return (int)hipGetLastError();
}

int launch_ctrl_occupancy(hipStream_t s, const GridView& g, int B, int N, const double* ctrl,
                          uint8_t* out_pt, uint8_t* out_line) {
    const int64_t total = (int64_t)B * N;
    if (total <= 0) return hipSuccess;
    const int block = 256;
    hipLaunchKernelGGL(k_ctrl_occupancy, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0, s, g, B, N,
                       ctrl, out_pt, out_line);
    return (int)hipGetLastError();
}

}  // namespace vigo

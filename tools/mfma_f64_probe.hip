// mfma_f64_probe.hip — dev tool: what v_mfma_f64_4x4x4_4b_f64 does to the 64 lanes of a wave on gfx950, measured, so that a
// wave-level fp64 sum can be built from it with a KNOWN order of additions:
//   1. which lanes of B (with A = 1 everywhere) and of A (with B = 1 everywhere) each lane of D sums  -> 64x64 incidence
//   2. the order in which the four products of one dot product are accumulated (values whose sum depends on it)
//   3. the latency of a dependent chain of such instructions, next to the DPP butterfly level it would replace
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_f64_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void __launch_bounds__(64) k_incidence(double* out_b, double* out_a) {
    const int l = threadIdx.x;
    for (int m = 0; m < 64; ++m) {
        const double onehot = l == m ? 1.0 : 0.0;
        double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, onehot, 0.0, 0, 0, 0);   // A = ones, B = e_m
        double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(onehot, 1.0, 0.0, 0, 0, 0);   // A = e_m, B = ones
        out_b[m * 64 + l] = d1;
        out_a[m * 64 + l] = d2;
    }
}

// A = ones, B = per-lane values: D per lane
__global__ void __launch_bounds__(64) k_sum(const double* b, double* out, double* out2) {
    const int l = threadIdx.x;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, b[l], 0.0, 0, 0, 0);
    out[l] = d;
    // second level: the first result goes in as A (transposed role), B = ones
    out2[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(d, 1.0, 0.0, 0, 0, 0);
}

__device__ __forceinline__ double dpp_f64_qp1(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

#define REP 1000
template <int MODE>
__global__ void __launch_bounds__(64) k_lat(double* out, long long* cyc, double seed) {
    double a = seed + threadIdx.x * 1e-3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        if (MODE == 0) {          // 4 dependent MFMA (result fed back as B)
#pragma unroll
            for (int j = 0; j < 4; ++j) a = __builtin_amdgcn_mfma_f64_4x4x4f64(0.25, a, 0.0, 0, 0, 0);
        } else if (MODE == 1) {   // 4 dependent (MFMA -> v_add_f64 -> MFMA ...): the hazard between the two units both ways
#pragma unroll
            for (int j = 0; j < 4; ++j) a = __builtin_amdgcn_mfma_f64_4x4x4f64(0.25, a, 0.0, 0, 0, 0) + 1e-9;
        } else if (MODE == 2) {   // 4 dependent butterfly levels (2 dpp + add)
#pragma unroll
            for (int j = 0; j < 4; ++j) a = (a + dpp_f64_qp1(a)) * 0.5;
        } else if (MODE == 3) {   // result fed back as A
#pragma unroll
            for (int j = 0; j < 4; ++j) a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 0.25, 0.0, 0, 0, 0);
        } else if (MODE == 4) {   // 4 dependent 16x16x4 (result register 0 fed back as B)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef double d4 __attribute__((ext_vector_type(4)));
                d4 z = {0, 0, 0, 0};
                d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(0.25, a, z, 0, 0, 0);
                a = r[0];
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void lat(const char* name, int per_iter) {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: report raw and per element
    printf("{\"chain\": \"%s\", \"counter_ticks\": %lld, \"ticks_per_element\": %.4f}\n", name, c, (double)c / (REP * per_iter));
    hipFree(out); hipFree(cyc);
}

int main() {
    double *ob, *oa;
    hipMalloc(&ob, 64 * 64 * 8); hipMalloc(&oa, 64 * 64 * 8);
    hipLaunchKernelGGL(k_incidence, dim3(1), dim3(64), 0, 0, ob, oa);
    std::vector<double> hb(4096), ha(4096);
    hipMemcpy(hb.data(), ob, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ha.data(), oa, 4096 * 8, hipMemcpyDeviceToHost);
    // for each D lane: the B lanes it sums (A = ones), the A lanes it sums (B = ones)
    printf("D lane <- B lanes (A = 1) | A lanes (B = 1)\n");
    for (int l = 0; l < 64; ++l) {
        printf("%2d <- B:", l);
        for (int m = 0; m < 64; ++m) if (hb[m * 64 + l] != 0.0) printf(" %d", m);
        printf(" | A:");
        for (int m = 0; m < 64; ++m) if (ha[m * 64 + l] != 0.0) printf(" %d", m);
        printf("\n");
    }
    // accumulation order: values with cancellation so that every order of the four additions gives another double
    srand(7);
    std::vector<double> b(64), d(64), d2(64);
    int votes[3] = {0, 0, 0}, trials = 0, votes2[3] = {0, 0, 0};
    double *db, *dd, *dd2;
    hipMalloc(&db, 512); hipMalloc(&dd, 512); hipMalloc(&dd2, 512);
    for (int t = 0; t < 200; ++t) {
        for (int l = 0; l < 64; ++l) b[l] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20);
        hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, 0, db, dd, dd2);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        hipMemcpy(d2.data(), dd2, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            // contributing B lanes of D lane l, ascending
            int src[4], n = 0;
            for (int m = 0; m < 64 && n < 4; ++m) if (hb[m * 64 + l] != 0.0) src[n++] = m;
            if (n != 4) continue;
            volatile double up = 0.0; for (int k = 0; k < 4; ++k) up = up + b[src[k]];
            volatile double dn = 0.0; for (int k = 3; k >= 0; --k) dn = dn + b[src[k]];
            volatile double p0 = b[src[0]] + b[src[1]], p1 = b[src[2]] + b[src[3]]; volatile double tr = p0 + p1;
            ++trials;
            votes[0] += d[l] == up; votes[1] += d[l] == dn; votes[2] += d[l] == tr;
            // second level: A lanes of D lane l
            int sa[4]; n = 0;
            for (int m = 0; m < 64 && n < 4; ++m) if (ha[m * 64 + l] != 0.0) sa[n++] = m;
            volatile double u2 = 0.0; for (int k = 0; k < 4; ++k) u2 = u2 + d[sa[k]];
            volatile double n2 = 0.0; for (int k = 3; k >= 0; --k) n2 = n2 + d[sa[k]];
            volatile double q0 = d[sa[0]] + d[sa[1]], q1 = d[sa[2]] + d[sa[3]]; volatile double t2 = q0 + q1;
            votes2[0] += d2[l] == u2; votes2[1] += d2[l] == n2; votes2[2] += d2[l] == t2;
        }
    }
    printf("{\"accumulation_order_level1\": {\"trials\": %d, \"ascending_lane\": %d, \"descending_lane\": %d, \"pairwise\": %d}}\n", trials, votes[0], votes[1], votes[2]);
    printf("{\"accumulation_order_level2\": {\"trials\": %d, \"ascending_lane\": %d, \"descending_lane\": %d, \"pairwise\": %d}}\n", trials, votes2[0], votes2[1], votes2[2]);
    lat<0>("4x4x4 mfma -> mfma (as B)", 4);
    lat<3>("4x4x4 mfma -> mfma (as A)", 4);
    lat<1>("4x4x4 mfma -> v_add_f64 -> mfma", 4);
    lat<2>("butterfly level (2 dpp + add) + mul", 4);
    lat<4>("16x16x4 mfma -> mfma (as B)", 4);
    lat<2>("butterfly level (2 dpp + add) + mul, again", 4);
    return 0;
}

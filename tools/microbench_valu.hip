// microbench_valu.hip — dev tool: issue cost (cycles per instruction, one wave per SIMD) of the
// instruction patterns k_optimize's critical path is made of, measured with s_memtime on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_valu.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 2000

__device__ __forceinline__ double dpp_f64_qp1(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, double seed) {
    double a = seed + threadIdx.x, b = seed * 0.5, c = seed * 0.25, d = seed * 0.125;
    double e = a + 1, f = b + 1, g = c + 1, h = d + 1;
    const double m = 1.0000001, ad = 1e-9;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; ++i) {
        if (MODE == 0) {  // 8 dependent fma
#pragma unroll
            for (int j = 0; j < 8; ++j) a = __builtin_fma(a, m, ad);
        } else if (MODE == 1) {  // 8 fma in 4 independent chains
#pragma unroll
            for (int j = 0; j < 2; ++j) { a = __builtin_fma(a, m, ad); b = __builtin_fma(b, m, ad); c = __builtin_fma(c, m, ad); d = __builtin_fma(d, m, ad); }
        } else if (MODE == 2) {  // 8 dependent add
#pragma unroll
            for (int j = 0; j < 8; ++j) a = a + ad;
        } else if (MODE == 3) {  // 8 independent adds (8 chains)
            a += ad; b += ad; c += ad; d += ad; e += ad; f += ad; g += ad; h += ad;
        } else if (MODE == 4) {  // butterfly level x4 dependent: (2 dpp + add) x 4  = 12 instr
#pragma unroll
            for (int j = 0; j < 4; ++j) a += dpp_f64_qp1(a);
        } else if (MODE == 5) {  // butterfly level on 4 independent values
            a += dpp_f64_qp1(a); b += dpp_f64_qp1(b); c += dpp_f64_qp1(c); d += dpp_f64_qp1(d);
        } else if (MODE == 6) {  // fp64 division chain x2
            a = b / a; a = c / a;
        } else if (MODE == 7) {  // 8 dependent fp32 fma
            float x = (float)a;
#pragma unroll
            for (int j = 0; j < 8; ++j) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
            a = x;
        } else if (MODE == 8) {  // 8 dependent mul
#pragma unroll
            for (int j = 0; j < 8; ++j) a = a * m;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ninstr, int blocks) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    // s_memtime counts at a fixed 100 MHz on gfx9; report wall ns per instruction from the event
    std::printf("%-44s blocks=%5d  %.3f ms  -> %.2f ns/instr (x2.4 GHz = %.1f cycles)  counter/instr %.3f\n", name, blocks, ms,
                ms * 1e6 / ((double)REP * ninstr), ms * 1e6 / ((double)REP * ninstr) * 2.4, (double)h[0] / ((double)REP * ninstr));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {256, 1024, 2048}) {
        run<0>("8 dependent v_fma_f64", 8, blocks);
        run<1>("8 v_fma_f64, 4 chains", 8, blocks);
        run<2>("8 dependent v_add_f64", 8, blocks);
        run<3>("8 independent v_add_f64", 8, blocks);
        run<8>("8 dependent v_mul_f64", 8, blocks);
        run<4>("4 dependent (2 dpp + add_f64) = 12 instr", 12, blocks);
        run<5>("4 independent (2 dpp + add_f64) = 12 instr", 12, blocks);
        run<6>("2 dependent fp64 divisions (count 2)", 2, blocks);
        run<7>("8 dependent v_fma_f32", 8, blocks);
    }
    return 0;
}

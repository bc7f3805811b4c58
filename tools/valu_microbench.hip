// Microbenchmark: what one SIMD of gfx950 sustains for wave64 fp64 / fp32 / mov / DPP streams at
// 1, 2 and 4 waves per SIMD.  Used to calibrate the fp64-VALU roofline quoted in DESIGN.md 4 and EXPERIMENTS.md 4.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *clk, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double m = 1.0000001, c = 1e-9;
    const float mf = 1.0000001f, cf = 1e-9f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // 8 independent fp64 FMA chains
                a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
                a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
            } else if (MODE == 1) {  // fp32 FMA
                f0 = fmaf(f0, mf, cf); f1 = fmaf(f1, mf, cf); f2 = fmaf(f2, mf, cf); f3 = fmaf(f3, mf, cf);
                f4 = fmaf(f4, mf, cf); f5 = fmaf(f5, mf, cf); f6 = fmaf(f6, mf, cf); f7 = fmaf(f7, mf, cf);
            } else if (MODE == 2) {  // fp64 add
                a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
            } else if (MODE == 3) {  // one dependent fp64 FMA chain
                a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
                a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
            } else if (MODE == 4) {  // DPP 32-bit moves feeding adds (fp32)
                f0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f0), 0xB1, 0xf, 0xf, false));
                f1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f1), 0x4E, 0xf, 0xf, false));
                f2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f2), 0x141, 0xf, 0xf, false));
                f3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f3), 0x140, 0xf, 0xf, false));
                f4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f4), 0x138, 0xf, 0xf, false));
                f5 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f5), 0x130, 0xf, 0xf, false));
                f6 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f6), 0xB1, 0xf, 0xf, false));
                f7 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f7), 0x4E, 0xf, 0xf, false));
            } else if (MODE == 5) {  // fp64 mul (v_mul_f64)
                a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const int gid = blockIdx.x * 64 + threadIdx.x;
    out[gid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
int run(const char *name, int waves) {
    double *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(double) * 64 * waves));
    CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * waves));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, clk, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, clk, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * waves);
    CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < waves; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    cyc /= waves; real /= waves;
    const double ninstr = (double)ITERS * 64;
    printf("%-22s waves/SIMD %4.1f  wall %.3f ms  shader-cycles/wave %.0f  cycles/instr(per wave) %.2f  clock %.2f GHz  "
           "chip instr/cycle/SIMD %.3f\n", name, waves / 1024.0, ms, cyc, cyc / ninstr, cyc / (real * 10.0) ,
           ninstr * waves / 1024.0 / (ms * 1e-3 * (cyc / (real * 10.0)) * 1e9));
    (void)hipFree(out); (void)hipFree(clk);
    return 0;
}

int main() {
    for (int w : {1024, 2048, 4096, 8192}) {
        run<0>("fp64 fma x8 indep", w); run<2>("fp64 add x8 indep", w); run<5>("fp64 mul x8 indep", w);
        run<3>("fp64 fma dependent", w); run<1>("fp32 fma x8 indep", w); run<4>("dpp mov + fp32 add", w);
        printf("\n");
    }
    return 0;
}

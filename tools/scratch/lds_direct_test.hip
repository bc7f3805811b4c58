// does global_load_lds_dwordx4 put lane i's 16 bytes at M0 + 16 i (masked lanes writing nothing)?  (gfx950; tools/scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double *g, double *o) {
    __shared__ double buf[18 * 72 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 18 * 72 + 64; i += 64) buf[i] = -1.0;
    __syncthreads();
    if (lane < 36) {
        const double *row = g + lane * 36;
#pragma unroll
        for (int kk = 0; kk < 18; ++kk)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row + 2 * kk),
                                             (__attribute__((address_space(3))) void *)(buf + kk * 72), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 18 * 72; i += 64) o[i] = buf[i];
}
int main() {
    std::vector<double> h(36 * 36);
    for (int i = 0; i < 36 * 36; ++i) h[i] = i;
    double *g, *o;
    hipMalloc(&g, sizeof(double) * 36 * 36); hipMalloc(&o, sizeof(double) * 18 * 72);
    hipMemcpy(g, h.data(), sizeof(double) * 36 * 36, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    std::vector<double> r(18 * 72);
    hipMemcpy(r.data(), o, sizeof(double) * 18 * 72, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int kk = 0; kk < 18; ++kk) for (int lane = 0; lane < 36; ++lane) for (int e = 0; e < 2; ++e) {
        const double want = lane * 36 + 2 * kk + e, got = r[kk * 72 + lane * 2 + e];
        if (want != got) { if (bad < 5) printf("mismatch k %d lane %d e %d: want %g got %g\n", kk, lane, e, want, got); ++bad; }
    }
    printf("lds-direct layout check: %d mismatches of %d\n", bad, 18 * 36 * 2);
    return bad != 0;
}

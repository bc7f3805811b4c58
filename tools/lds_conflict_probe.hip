// LDS bank-conflict probe for the access patterns of the Riccati kernel (ik_ddp.hip, BackwardLds): one dispatch per pattern,
// one wave, every lane repeating ONE LDS instruction at its own byte address.  Run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
// and read the counters per dispatch (the program prints dispatch order = pattern names).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/scratch/lds_conflict_probe tools/lds_conflict_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

enum Op { RD64 = 0, RD2_64 = 1, WR64 = 2, RD128 = 3 };

template <int OP>
__global__ void __launch_bounds__(64) probe(const unsigned *addr, int iters, double *sink) {
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;
    __syncthreads();
    const unsigned a = addr[threadIdx.x] + (unsigned)(unsigned long long)lds;
    double acc = 0.0;
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (int i = 0; i < iters; ++i) {
        if (OP == RD64) { double v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc += v; }
        if (OP == RD2_64) { d2 v; asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc += v.x + v.y; }
        if (OP == WR64) { asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(a), "v"(acc) : "memory"); acc += 1.0; }
        if (OP == RD128) { d2 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc += v.x + v.y; }
    }
    sink[threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const int iters = 4096;
    unsigned *d_addr; double *d_sink;
    hipMalloc(&d_addr, 64 * sizeof(unsigned)); hipMalloc(&d_sink, 64 * sizeof(double));
    struct Pat { std::string name; int op; std::function<int(int)> elem; bool active_all; };   // elem(lane) = double index, -1 = lane idle (takes lane 0's)
    const int LD = 37, LDK = 21;
    auto row_r = [](int lane) { return lane < 36 ? lane : (lane < 54 ? lane - 18 : 0); };
    std::vector<Pat> pats;
    for (int ld : {37, 36, 38, 40}) {
        pats.push_back({"row read2 (lane r: N[r*" + std::to_string(ld) + "+j]) j=0", RD2_64, [=](int l) { return row_r(l) * ld; }, true});
        pats.push_back({"row read2 ld" + std::to_string(ld) + " j=10", RD2_64, [=](int l) { return row_r(l) * ld + 10; }, true});
        pats.push_back({"row write b64 ld" + std::to_string(ld), WR64, [=](int l) { return row_r(l) * ld + 3; }, true});
        pats.push_back({"col read b64 (lane r: N[j*ld+r]) ld" + std::to_string(ld), RD64, [=](int l) { return 5 * ld + row_r(l); }, true});
        pats.push_back({"col write b64 ld" + std::to_string(ld), WR64, [=](int l) { return 7 * ld + row_r(l); }, true});
        pats.push_back({"mfma acc read (N[(lk+4v)*ld + li]) ld" + std::to_string(ld), RD64, [=](int l) { return (l >> 4) * ld + (l & 15); }, true});
        pats.push_back({"mfma acc write ld" + std::to_string(ld), WR64, [=](int l) { return (l >> 4) * ld + (l & 15); }, true});
        pats.push_back({"mfma mirror write (N[li*ld + lk]) ld" + std::to_string(ld), WR64, [=](int l) { return (l & 15) * ld + (l >> 4); }, true});
    }
    for (int ldk : {21, 20, 22, 24}) {
        pats.push_back({"mfma operand read (Ys[li*ldk + lk]) ldk" + std::to_string(ldk), RD64, [=](int l) { return (l & 15) * ldk + (l >> 4); }, true});
        pats.push_back({"Ys row write (Ys[r*ldk+p]) ldk" + std::to_string(ldk), WR64, [=](int l) { return row_r(l) * ldk + 5; }, true});
    }
    pats.push_back({"broadcast b128", RD128, [](int) { return 64; }, true});
    pats.push_back({"broadcast read2", RD2_64, [](int) { return 64; }, true});
    pats.push_back({"Lc packed write q=0 (lane 36+p: p(p-1)/2)", WR64, [](int l) { int p = l - 36; return (p >= 1 && p < 18) ? p * (p - 1) / 2 : 4000 + l; }, true});
    pats.push_back({"contiguous b64 read", RD64, [](int l) { return l; }, true});
    pats.push_back({"contiguous b128 read", RD128, [](int l) { return 2 * l; }, true});
    int n = 0;
    for (auto &p : pats) {
        unsigned h[64];
        for (int l = 0; l < 64; ++l) h[l] = 8u * (unsigned)p.elem(l);
        hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
        switch (p.op) {
            case RD64: hipLaunchKernelGGL(probe<RD64>, dim3(1), dim3(64), 0, 0, d_addr, iters, d_sink); break;
            case RD2_64: hipLaunchKernelGGL(probe<RD2_64>, dim3(1), dim3(64), 0, 0, d_addr, iters, d_sink); break;
            case WR64: hipLaunchKernelGGL(probe<WR64>, dim3(1), dim3(64), 0, 0, d_addr, iters, d_sink); break;
            default: hipLaunchKernelGGL(probe<RD128>, dim3(1), dim3(64), 0, 0, d_addr, iters, d_sink); break;
        }
        hipDeviceSynchronize();
        printf("dispatch %d: %s\n", ++n, p.name.c_str());
    }
    return 0;
}

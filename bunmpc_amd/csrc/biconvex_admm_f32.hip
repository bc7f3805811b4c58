// The fp32 instantiations of the batched centroidal ADMM kernel (BASELINE config 3: fp32 iterates, operators and projections;
// every decision of the algorithm reduced and compared in fp64 -- see biconvex_admm.hip for the mapping and the reference lines).
//
// A translation unit of its own because it is built with -fno-slp-vectorize (bunmpc_amd/build.py): hipcc's SLP vectoriser packs
// pairs of fp32 operations into v_pk_mul/fma/add_f32, which need their operands in adjacent register pairs -- 47 v_mov_b32 per
// backtracking step to put them there, duplicated copies of the broadcast constants, 284-307 registers, and so 40-60 values
// spilled to scratch memory under the cap of 256 that two waves per SIMD need (345 MB of HBM traffic per launch against 75 MB of
// inputs and results; profiles/r02_pmc_hbm_cfg3.txt).  Without the packing the body takes 200-212 registers: no scratch, and
// Go2 H = 40, B = 4096 goes 7.09 -> 6.02 ms.  (The fp64 kernels are ~0.5 % faster WITH the vectoriser, hence the split.)
#include "biconvex_kernels.h"

namespace bunmpc {
namespace {

#include "biconvex_lanes.h"
#include "biconvex_admm_body.h"

// TWO waves per SIMD: this latency-bound loop gains a second wave to issue from while the first waits (one wave per SIMD: 9.0 ms)
template <int LPP, int E>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void biconvex_admm_kernel_f32(const BatchArgs a) {
    admm_body<float, LPP, E, false, false>(a);
}

}  // namespace

hipError_t launch_biconvex_admm_f32(const BatchArgs &a, int lpp, unsigned grid, size_t lds, hipStream_t stream) {
    if (lpp == 16) hipLaunchKernelGGL((biconvex_admm_kernel_f32<16, 4>), dim3(grid), dim3(64), lds, stream, a);
    else if (lpp == 32) hipLaunchKernelGGL((biconvex_admm_kernel_f32<32, 4>), dim3(grid), dim3(64), lds, stream, a);
    else if (lpp == 64) hipLaunchKernelGGL((biconvex_admm_kernel_f32<64, 4>), dim3(grid), dim3(64), lds, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// private-segment (scratch) bytes per lane of the fp32 instantiations, the largest of the three: 0 is the point of this file
int biconvex_admm_f32_scratch_bytes() {
    size_t worst = 0;
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&biconvex_admm_kernel_f32<16, 4>)) != hipSuccess) return -1;
    worst = at.localSizeBytes > worst ? at.localSizeBytes : worst;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&biconvex_admm_kernel_f32<32, 4>)) != hipSuccess) return -1;
    worst = at.localSizeBytes > worst ? at.localSizeBytes : worst;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&biconvex_admm_kernel_f32<64, 4>)) != hipSuccess) return -1;
    worst = at.localSizeBytes > worst ? at.localSizeBytes : worst;
    return (int)worst;
}

}  // namespace bunmpc

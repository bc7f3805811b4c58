// Batched centroidal bi-convex ADMM (force-QP / motion-QP alternation by projected
// FISTA) for gfx950 (MI355X, CDNA4).  One launch = B independent
// BiConvexMP::optimize(x_init, num_iters) calls, every ADMM and FISTA iteration
// inside the kernel.
//
// Reference behaviour restated (paths under iterative_supervised_learning/):
//   src/motion_planner/biconvex.cpp:80-120   ADMM loop, P update, exits
//   src/motion_planner/biconvex.cpp:27-78    create_bound_constraints / create_cost_X / _F
//   src/dynamics/centroidal.cpp:57-84        A_x, b_x (force step)
//   src/dynamics/centroidal.cpp:6-37,86-127  A_f, b_f (motion step)
//   include/dynamics/centroidal.hpp:22-27    x_init rows
//   src/solvers/problem.cpp:31-56            gradient / objective difference
//   src/solvers/fista.cpp:6-70               FISTA, backtracking, "SoC" projection
//
// MI355X mapping (this is not how the reference is organised):
//   * one knot per lane, one problem per LPP-lane segment of a wave64
//     (LPP = 16/32/64 >= H+1), so a wave carries 4/2/1 problems; 64-thread
//     workgroups, B*LPP/64 of them -- one barrier (momentum table), no inter-workgroup traffic;
//   * matrix-free operators: lane t applies its own 6x12 block of A_x and its own
//     block-row / block-column of the block-bidiagonal A_f; the explicit Hessian
//     2(Q + rho A^T A) the reference rebuilds every ADMM iteration never exists;
//   * knot t <-> t+-1 coupling of A_f through DPP wave shifts (v_mov_b32_dpp
//     wave_shr/wave_shl), the three per-iteration scalars (||d||^2, g.d, objective
//     difference) through a DPP butterfly + v_permlane16/32_swap -- all lanes of a
//     segment end up with bit-identical sums, so every accept / exit decision is
//     segment-uniform without a broadcast;
//   * the affine images A y + bPk are carried through the momentum step by
//     linearity, so an iteration costs one A and one A^T application instead of the
//     reference's three sparse mat-vecs;
//   * fp64 arithmetic (R = double; MFMA has no advantage over VALU for fp64 on gfx950 and the
//     blocks are 6x12 / 9x9 sparse), FISTA state in VGPRs; between phases X / F / P of a problem
//     rest in LDS (each lane touches only its own knot's blocks), so HBM sees the inputs once and
//     the results once;
//   * R = float is the mixed-precision variant (BASELINE config 3): iterates, operators and
//     projections in fp32, while every decision the algorithm takes -- the backtracking test,
//     both exit tests, the dynamics violation -- is reduced and compared in fp64.  HBM keeps fp64.
#include "biconvex_kernels.h"
#include <algorithm>
#include <mutex>

namespace bunmpc {
namespace {

#include "biconvex_lanes.h"

#include "biconvex_admm_body.h"

// fp64, WPE = 1: ONE wave per SIMD.  The body holds 294 registers; capped at 256 with the FISTA iterates in registers the compiler parks
// 63 of them in scratch memory, some inside the loops (round 2 measured that build 4 % faster at 14 x the HBM traffic, round 3 level).
// WPE = 2 (round 4): two waves per SIMD with x_k and its image in LDS (biconvex_admm_body.h: XLDS) -- both FISTA loops free of
// scratch accesses at 256 registers, the two waves of a SIMD covering each other's latencies.  A lone wave of this build is slower
// than a lone wave of the other (2.30 against 1.94 ms: the LDS round trip sits on its chain), so it is taken where the batch
// needs more waves than the chip has SIMDs and three problems per wave do not save a round (launch_biconvex_admm below):
// B = 4096, H = 20: 3.74 against 4.02 ms.
template <typename R, int LPP, int E, bool RAW, bool HASQF, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void biconvex_admm_kernel(const BatchArgs a) {
    admm_body<R, LPP, E, RAW, HASQF, false, WPE == 2>(a);
}
// horizons of 64 .. 255 knots: one problem per workgroup of WAVES waves (biconvex_admm_body.h: WAVES)
template <int WAVES, bool RAW, bool HASQF, int WPE>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void biconvex_admm_wg_kernel(const BatchArgs a) {
    admm_body<double, 64, 4, RAW, HASQF, false, WPE == 2, WAVES>(a);
}
// the work-stealing variant (biconvex_admm_body.h: STEAL): three problems per wave, harness form, fp64
template <int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void biconvex_admm_steal_kernel(const BatchArgs a) {
    admm_body<double, 21, 4, false, false, true, WPE == 2>(a);
}
__global__ __launch_bounds__(64) void lane_selftest_kernel(const double *in, double *out) {
    const int i = threadIdx.x;
    const double v = in[i];
    out[i] = from_prev(v);
    out[64 + i] = from_next(v);
    out[128 + i] = seg_sum<16>(v);
    out[192 + i] = seg_sum<32>(v);
    out[256 + i] = seg_sum<64>(v);
    out[320 + i] = (double)__popcll(__ballot(v > 0.0));
    out[384 + i] = seg_sum<21>(i < 63 ? v : 0.0);                                                    // every lane of a 21-lane segment
    double a = i < 63 ? v : 0.0, b = i < 63 ? 2.0 * v : 0.0;
    seg_sum2<21>(a, b);                                                                                // the designated lanes only
    out[448 + i] = a;
    out[512 + i] = b;
    out[576 + i] = (double)seg_uniform<21>(__ballot(v > 40.0) & seg_desig<21>());                  // the spread masks, as a number (< 2^63: exact up to 2^53 -- compared by bits on the host through two halves)
    out[640 + i] = (double)(unsigned)(seg_uniform<21>(__ballot(i == 16 || i == 48)) >> 32);
    out[704 + i] = (double)(unsigned)(seg_uniform<21>(__ballot(i == 16 || i == 48)) & 0xffffffffu);
}

template <typename R, int LPP, bool RAW, bool HASQF>
hipError_t launch(const BatchArgs &a, bool two_per_simd, hipStream_t stream) {
    const int per_wave = 64 / LPP;
    const unsigned grid = (unsigned)((a.B + per_wave - 1) / per_wave);
    const size_t nstate = (size_t)kSegLds + (size_t)kKnotLds * (size_t)(a.H + 1);   // X, P, F, R of one problem
    const size_t lds = sizeof(R) * (kLdsZeros + per_wave * nstate);
    if (sizeof(R) == sizeof(float)) return launch_biconvex_admm_f32(a, LPP, grid, lds, stream);      // biconvex_admm_f32.hip
    if (two_per_simd) hipLaunchKernelGGL((biconvex_admm_kernel<double, LPP, 4, RAW, HASQF, 2>), dim3(grid), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((biconvex_admm_kernel<double, LPP, 4, RAW, HASQF, 1>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

template <int WAVES, bool RAW, bool HASQF, int WPE>
hipError_t launch_wg(const BatchArgs &a, hipStream_t stream) {
    const size_t lds = sizeof(double) * (kLdsZeros + (size_t)kSegLds + (size_t)kKnotLds * (size_t)(a.H + 1) + (size_t)WAVES * 40);
    static std::once_flag once;      // (more than the 64 KB a kernel may take without asking, from 209 knots on)
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] { attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&biconvex_admm_wg_kernel<WAVES, RAW, HASQF, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); });
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((biconvex_admm_wg_kernel<WAVES, RAW, HASQF, WPE>), dim3((unsigned)a.B), dim3(64 * WAVES), lds, stream, a);
    return hipGetLastError();
}
template <int WAVES, int WPE>
hipError_t launch_wg_form(const BatchArgs &a, hipStream_t stream) {
    if (a.precision != 0) return hipErrorInvalidValue;      // (fp64 only)
    if (!a.raw) return launch_wg<WAVES, false, false, WPE>(a, stream);
    return a.qf ? launch_wg<WAVES, true, true, WPE>(a, stream) : launch_wg<WAVES, true, false, WPE>(a, stream);
}

// FISTA's momentum coefficients: t+ = 1 + sqrt(1 + 4 t^2)/2 (sic, fista.cpp:34), c_k = (t_k - 1)/t_{k+1} -- a function of k alone, so
// one table per device, filled once by this kernel (until round 4 every wave tabulated them in its LDS: 1.2 KB of the 20 KB a wave
// may hold when eight of them share a CU)
__global__ void momentum_table_kernel(double *tab, int n) {
    double tk = 1.0;
    for (int i = 0; i < n; ++i) {
        const double tk1 = 1.0 + sqrt(1.0 + 4.0 * tk * tk) * 0.5;
        tab[i] = (tk - 1.0) / tk1;
        tk = tk1;
    }
}
const double *momentum_table(hipStream_t stream) {
    static std::mutex lock;
    static double *tab[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> hold(lock);
    if (!tab[dev]) {
        double *t = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&t), kMaxFistaIters * sizeof(double)) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(momentum_table_kernel, dim3(1), dim3(1), 0, stream, t, kMaxFistaIters);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(t); return nullptr; }      // (once per device: later launches on any stream find it filled)
        tab[dev] = t;
    }
    return tab[dev];
}

// Device counters of the work-stealing launches: a ring of 64 per device, one per launch in flight (a launch zeroes its own on its
// stream in front of the kernel; with 64 a counter comes round again only after 63 later launches of this process)
int *steal_counter(hipStream_t stream) {
    static std::mutex lock;
    static int *ring[16] = {};
    static unsigned next = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> hold(lock);
    if (!ring[dev] && hipMalloc(reinterpret_cast<void **>(&ring[dev]), 64 * sizeof(int)) != hipSuccess) { ring[dev] = nullptr; return nullptr; }
    int *c = ring[dev] + (next++ % 64);
    if (hipMemsetAsync(c, 0, sizeof(int), stream) != hipSuccess) return nullptr;
    return c;
}
// the persistent grid of the work-stealing kernel: as many waves as the chip holds at once (one or two per SIMD)
int g_steal_grid = 0;      // waves of the persistent grid (experiments, set_steal_grid below): 0 = one or two per SIMD
hipError_t launch_steal(const BatchArgs &a, long simds, bool two_per_simd, hipStream_t stream) {
    BatchArgs s = a;
    s.queue = steal_counter(stream);
    if (!s.queue) return hipErrorOutOfMemory;
    const size_t nstate = (size_t)kSegLds + (size_t)kKnotLds * (size_t)(a.H + 1);
    const size_t lds = sizeof(double) * (kLdsZeros + 3 * nstate);
    const long waves = g_steal_grid > 0 ? std::min<long>(g_steal_grid, (a.B + 2) / 3) : (two_per_simd ? 2 * simds : simds);
    if (two_per_simd) hipLaunchKernelGGL(biconvex_admm_steal_kernel<2>, dim3((unsigned)waves), dim3(64), lds, stream, s);
    else hipLaunchKernelGGL(biconvex_admm_steal_kernel<1>, dim3((unsigned)waves), dim3(64), lds, stream, s);
    return hipGetLastError();
}

template <int LPP>
hipError_t launch_lpp(const BatchArgs &a, bool two_per_simd, hipStream_t stream) {
    if (a.precision == 1) {   // fp32 arithmetic: harness form only
        if (a.raw || LPP == 21) return hipErrorInvalidValue;
        return launch<float, LPP == 21 ? 32 : LPP, false, false>(a, false, stream);
    }
    if (!a.raw) return launch<double, LPP, false, false>(a, two_per_simd, stream);
    return a.qf ? launch<double, LPP, true, true>(a, two_per_simd, stream) : launch<double, LPP, true, false>(a, two_per_simd, stream);
}

}  // namespace

// Horizons of 17..21 knots: 0 = 32-lane segments (two problems per wave), 1 = 21-lane segments (three per wave), 2 (default) = whichever
// finishes the batch sooner.  The kernel runs one wave per SIMD; a wave of three problems takes ~8 % longer than a wave of two
// (the segment sums cost more), so three per wave wins whenever it needs fewer ROUNDS of waves over the chip's SIMDs -- at
// B = 4096 on an MI355X (1024 SIMDs) both need two rounds, 1366 waves or 2048, and two per wave is the faster one; at B = 3072 or
// 6144 three per wave saves a whole round (1.41e6 solves/s against 1.03e6) -- or when the waves' run times differ widely anyway
// (num_iters well above ten: the ADMM's early exit, biconvex.cpp:111-114, makes the iteration counts differ per problem and the
// scheduler backfills; measured at num_iters = 100, B = 4096: 31 -> 28 ms).  (Tried and dropped: B = 4096 as one round of three per
// wave for 3072 problems + the one-problem-per-wave kernel for the other 1024 -- 2.29 + 1.7 ms, level with 2 x 2.02 ms.)
int set_steal_grid(int waves) { const int old = g_steal_grid; g_steal_grid = waves; return old; }
static int g_three_per_wave = 2;
int set_three_per_wave(int on) { const int old = g_three_per_wave; g_three_per_wave = on; return old; }
static long chip_simds() {
    static int simds = 0;
    if (simds == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        simds = 4 * cus;
    }
    return simds;
}
static bool three_per_wave_pays(const BatchArgs &a) {
    if (g_three_per_wave != 2) return g_three_per_wave != 0;
    const long simds = chip_simds(), w3 = (a.B + 2) / 3, w2 = (a.B + 1) / 2;
    return (w3 + simds - 1) / simds < (w2 + simds - 1) / simds || a.c.num_iters >= 25;
}
// Two waves per SIMD (the XLDS build): 0 never, 1 whenever the kernel exists for the shape (16 / 32 / 64 lanes per problem, fp64),
// 2 (default) where it is the faster one: the batch needs more waves than the chip has SIMDs.  Results do not depend on it.
static int g_two_per_simd = 2;
int set_two_waves_per_simd(int mode) { const int old = g_two_per_simd; g_two_per_simd = mode; return old; }
static bool two_per_simd_pays(const BatchArgs &a, int per_wave) {
    if (a.precision != 0 || g_two_per_simd == 0) return false;
    if (g_two_per_simd == 1) return true;
    return (a.B + per_wave - 1) / per_wave > chip_simds();
}
static thread_local int t_last_wpe = 1;
int biconvex_last_waves_per_simd() { return t_last_wpe; }
static int g_work_stealing = 1;       // 0: never the work-stealing kernel (tests: results must not depend on it)
int set_work_stealing(int on) { const int old = g_work_stealing; g_work_stealing = on; return old; }
static int g_latency_max_batch = 1024;
int set_latency_mapping_max_batch(int max_batch) { const int old = g_latency_max_batch; g_latency_max_batch = max_batch; return old; }
static int g_exact_step_decisions = 0;
int set_exact_step_decisions(int on) { const int old = g_exact_step_decisions; g_exact_step_decisions = on; return old; }
// which kernel the calling host thread's latest launch_biconvex_admm took (tests of the default dispatch; profiles)
static thread_local const char *t_last_kernel = "";
static thread_local int t_last_lpp = 0;       // lanes per problem of that launch (0: the one-problem-per-wave kernel)
int biconvex_last_lanes_per_problem() { return t_last_lpp; }
const char *biconvex_last_kernel_name() { return t_last_kernel; }

hipError_t launch_biconvex_admm(const BatchArgs &args, int n_eff, hipStream_t stream) {
    BatchArgs a = args;
    if (n_eff != 4 || a.H < 1 || a.H + 1 > kMaxKnots || a.B < 0 || (a.precision != 0 && a.precision != 1))
        return hipErrorInvalidValue;
    if (a.B == 0) return hipSuccess;
    if (a.c.maxit > kMaxFistaIters) return hipErrorInvalidValue;
    // the kernels address a wave's problems by 32-bit byte offsets from the wave's first problem (at most four problems)
    for (long stride : {a.sW_X, a.sW_X_ter, a.sW_F, a.sbounds})
        if (stride < 0 || stride > (1L << 26)) return hipErrorInvalidValue;
    // few problems, short horizon: one problem per wave (the chain of a solve is ~2.3x shorter; biconvex_latency.hip)
    if (a.B <= g_latency_max_batch && latency_mapping_fits(a, n_eff)) {
        t_last_kernel = "biconvex_latency_kernel";
        t_last_lpp = 0;
        BatchArgs al = a;
        al.exact_step_decisions = g_exact_step_decisions;
        return launch_biconvex_latency(al, stream);
    }
    a.cmtab = momentum_table(stream);
    if (!a.cmtab) return hipErrorOutOfMemory;
    const int k = a.H + 1;
    t_last_kernel = a.precision == 1 ? "biconvex_admm_kernel_f32" : "biconvex_admm_kernel";
    t_last_wpe = 1;
    if (k <= 16) { t_last_lpp = 16; const bool w2 = two_per_simd_pays(a, 4); t_last_wpe = w2 ? 2 : 1; return launch_lpp<16>(a, w2, stream); }
    // 17..21 knots (the headline shape): three problems per wave in 21-lane segments (fp64; the fp32 kernels keep 32-lane segments)
    if (k <= 21 && k > 16 && a.precision == 0 && three_per_wave_pays(a)) {
        t_last_lpp = 21;
        // many ADMM iterations (the early exit makes the counts differ per problem) and more waves than the chip holds: segments that
        // finish take the next problem (biconvex_admm_body.h: STEAL); the 32-bit offsets from the problem index must fit
        const long S = chip_simds(), per = std::max<long>({(long)a.H * 128, 9L * (a.H + 1) * 8, a.sW_X * 8, a.sW_F * 8, a.sbounds * 8, (long)a.c.num_iters * 16});
        if (g_work_stealing && !a.raw && a.c.num_iters >= 25 && (a.B + 2) / 3 > S && (double)a.B * (double)per < 2.0e9) {
            t_last_kernel = "biconvex_admm_steal_kernel";
            // (one wave per SIMD unless forced: measured at B = 4096, num_iters = 100: 30.7 ms; the two-waves build with grids of
            // 1024 .. 2048 waves 34.2 .. 37.3 ms -- the stealing itself already fills the gaps the second wave would)
            const bool w2 = g_two_per_simd == 1;
            t_last_wpe = w2 ? 2 : 1;
            return launch_steal(a, S, w2, stream);
        }
        const bool w2 = two_per_simd_pays(a, 3);
        t_last_wpe = w2 ? 2 : 1;
        return launch_lpp<21>(a, w2, stream);
    }
    if (k > 64) {      // 65 .. 256 knots: a workgroup of two, three or four waves per problem
        t_last_kernel = "biconvex_admm_wg_kernel";
        t_last_lpp = k <= 128 ? 128 : (k <= 192 ? 192 : 256);
        // the two-waves-per-SIMD build when there are more waves than SIMDs -- and, for two waves per problem, when four such workgroups'
        // LDS fits a CU (at 127 knots only three do: 9.2 ms against 6.9 at B = 1024); four waves per problem: always (11.4-12.6 ms
        // against 15.9-16.8: tools/horizon_sweep.py)
        const size_t lds_bytes = sizeof(double) * (kLdsZeros + (size_t)kSegLds + (size_t)kKnotLds * (size_t)k + (size_t)(t_last_lpp / 64) * 40);
        const bool fits = k > 128 || 4 * lds_bytes <= 160 * 1024;
        const bool w2 = g_two_per_simd == 1 || (g_two_per_simd == 2 && fits && (long)a.B * (t_last_lpp / 64) > chip_simds());
        t_last_wpe = w2 ? 2 : 1;
        if (k <= 128) return w2 ? launch_wg_form<2, 2>(a, stream) : launch_wg_form<2, 1>(a, stream);
        if (k <= 192) return w2 ? launch_wg_form<3, 2>(a, stream) : launch_wg_form<3, 1>(a, stream);
        return w2 ? launch_wg_form<4, 2>(a, stream) : launch_wg_form<4, 1>(a, stream);
    }
    t_last_lpp = k <= 32 ? 32 : 64;
    const bool w2 = two_per_simd_pays(a, 64 / t_last_lpp);
    t_last_wpe = w2 ? 2 : 1;
    if (k <= 32) return launch_lpp<32>(a, w2, stream);
    return launch_lpp<64>(a, w2, stream);
}

hipError_t launch_lane_selftest(const double *in, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(lane_selftest_kernel, dim3(1), dim3(64), 0, stream, in, out);
    return hipGetLastError();
}

const char *biconvex_kernel_name(int H, int raw) {
    (void)raw;
    const int k = H + 1;
    return k <= 16 ? "biconvex_admm_kernel<double, 16" : (k <= 21 && g_three_per_wave == 1 ? "biconvex_admm_kernel<double, 21" : (k <= 32 ? "biconvex_admm_kernel<double, 32" : "biconvex_admm_kernel<double, 64"));
}

}  // namespace bunmpc

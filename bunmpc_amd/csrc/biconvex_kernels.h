// Internal (not part of the C-ABI): argument block shared by the host launcher
// (bunmpc_capi.hip) and the batched centroidal ADMM kernel (biconvex_admm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace bunmpc {

// Solver constants (reference defaults: biconvex.hpp:146-160, fista.hpp:52-60).
struct SolverConsts {
    double m;         // robot mass
    double rho;       // ADMM penalty
    double mu;        // friction coefficient of the "SoC" projection
    double beta;      // backtracking growth
    double tol;       // FISTA exit: ||y+ - y|| < tol
    double exit_tol;  // ADMM exit: ||A_f X - b_f|| < exit_tol
    int maxit;        // FISTA iteration cap
    int num_iters;    // ADMM iteration cap
};

// One batch of B independent BiConvexMP::optimize calls.  All pointers are DEVICE
// pointers.  Layout (row-major, batch outermost, then knot, then component):
//   cnt_plan [B][H][E][4]   rows [flag, x, y, z]            (set_contact_plan)
//   dt       [B][H]
//   x_init   [B][9]
// cost / bound description, one of two forms:
//   harness form (raw == 0): what create_cost_X / create_cost_F /
//     create_bound_constraints receive; the kernel applies those formulas itself.
//       W_X [.][9H], W_X_ter [.][9], W_F [.][3EH], bounds [.][H][6]  (batch stride 0 = shared)
//       X_nom [B][9H], X_ter [B][9]
//   raw form (raw == 1): what set_cost_x / set_cost_f / set_bounds_x leave behind
//       Qx, qx, lbx, ubx [B][9(H+1)] (Qx = diagonal), Qf [B][3EH], qf [B][3EH] or null
// state, in = warm start (set_warm_start_vars), out = last iterates:
//   X [B][9(H+1)], F [B][3EH], P [B][9(H+1)], L_x [B], L_f [B]
// telemetry: dyn_viol [B] (last ||A_f X - b_f||), hist [B][num_iters] or null,
//   stats [B][6] = {admm iters, sum F-FISTA iters, sum X-FISTA iters, F retries,
//                   X retries, status (0 ok, 2 NaN)}
struct BatchArgs {
    int B, H, raw, cold_start;
    int precision;   // 0: fp64 arithmetic; 1: fp32 iterates with fp64 decisions (harness form only)
    int exact_step_decisions;   // 1: the one-problem-per-wave kernel skips the fp32 shortcut of its step decisions (tests)
    double L0_x, L0_f;
    SolverConsts c;
    const double *cnt_plan, *dt, *x_init;
    const double *W_X, *W_X_ter, *W_F, *bounds, *X_nom, *X_ter;
    long sW_X, sW_X_ter, sW_F, sbounds;
    const double *Qx, *qx, *lbx, *ubx, *Qf, *qf;
    double *X, *F, *P, *L_x, *L_f;
    double *dyn_viol, *hist;
    int *stats;
    int *trace;      // [B][num_iters][4] running totals {it_f, it_x, bt_f, bt_x} after every ADMM iteration, or null
    const double *cmtab;   // (set by the launcher) FISTA's momentum coefficients (t_k - 1) / t_{k+1}, k = 0 .. kMaxFistaIters - 1: a function of k alone
    int *queue;      // (set by the launcher) the work-stealing kernel's device counter: problems handed out beyond the first per segment
};

constexpr int kStats = 6;
constexpr int kKnotLds = 39;           // LDS elements per knot of a problem (biconvex_admm_body.h: X 9, P 9, F 12, R 9)
constexpr int kSegLds = 15;            // ... and per problem in front of its knots (the x_init rows' multipliers, 9; XLDS: step constants, violation, counters)
constexpr int kLdsZeros = 54;          // zeros in LDS in front of all that (lanes without a knot read them)
constexpr int kMaxFistaIters = 4096;  // length of the momentum table (one per device, momentum_table below; the one-problem-per-wave kernel keeps its own in LDS: 32 KB + <= 30 KB of iterates < 64 KB)
constexpr int kMaxKnots = 256; // H + 1 <= 256: one knot per lane, one problem per <= 64 lanes of a wave, or (65 .. 256 knots) per workgroup of 2 / 4 waves

// Launch the batched ADMM kernel on `stream`.  Returns hipSuccess or the launch error;
// hipErrorInvalidValue for unsupported shapes (n_eff != 4, H + 1 > 64).
hipError_t launch_biconvex_admm(const BatchArgs &a, int n_eff, hipStream_t stream);

// The one-problem-per-wave mapping (biconvex_latency.hip): fp64, n_eff = 4, H + 1 <= 21.  launch_biconvex_admm takes it for
// batches of at most latency_mapping_max_batch() problems that fit.
bool latency_mapping_fits(const BatchArgs &a, int n_eff);
hipError_t launch_biconvex_latency(const BatchArgs &a, hipStream_t stream);
int set_latency_mapping_max_batch(int max_batch);   // returns the old value
int set_three_per_wave(int mode);                    // 21-lane segments for 17..21 knots: 0 never, 1 always, 2 when it pays (default); returns the old value
int set_two_waves_per_simd(int mode);               // the two-waves-per-SIMD build of the fp64 batch kernel: 0 never, 1 always, 2 when it pays (default); returns the old value
int biconvex_last_waves_per_simd();                  // of the calling host thread's latest launch (1 or 2)
int set_steal_grid(int waves);                       // waves of the work-stealing kernel's persistent grid (experiments; 0 = what the chip holds); returns the old value
int set_work_stealing(int on);                       // the segment-level work-stealing kernel for num_iters >= 25 (default on); returns the old value
int biconvex_last_lanes_per_problem();               // of the calling host thread's latest launch: 16 / 21 / 32 / 64, 0 = one problem per wave
int set_exact_step_decisions(int on);                // ... takes every step decision from the fp64 sums; returns the old value

// fp32 instantiations (biconvex_admm_f32.hip); called by launch_biconvex_admm with the lanes per problem (16 / 32 / 64), the grid
// and the LDS bytes it has worked out
hipError_t launch_biconvex_admm_f32(const BatchArgs &a, int lpp, unsigned grid, size_t lds, hipStream_t stream);

int biconvex_admm_f32_scratch_bytes();    // private-segment bytes per lane of the fp32 kernels (hipFuncGetAttributes), -1 on error

// Lane-exchange self test (DPP shifts and segment sums used by the kernel).
// out must hold 12*64 doubles.
hipError_t launch_lane_selftest(const double *in, double *out, hipStream_t stream);

// Name of the kernel symbol for a given H (for profiling / bench reports).
const char *biconvex_kernel_name(int H, int raw);
// ... and of the kernel the calling host thread's latest launch_biconvex_admm actually took
const char *biconvex_last_kernel_name();

}  // namespace bunmpc

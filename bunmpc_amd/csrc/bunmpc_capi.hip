// C-ABI of libbunmpc_hip.so (declared in include/bunmpc.h): host-side handles that keep
// the reference's call semantics, and the launchers of the gfx950 kernels.
//
// Host classes mirror (behaviour, not code) iterative_supervised_learning/
//   include/motion_planner/biconvex.hpp + src/motion_planner/biconvex.cpp   (BiConvexMP)
//   src/dynamics/centroidal.cpp:39-49                                       (contact arrays)
//   src/gait_planner/gait_planner.cpp                                       (QuadrupedGait)
// All numerical work of optimize() happens in biconvex_admm.hip on the GPU.
#include "../../include/bunmpc.h"
#include "biconvex_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(BMPC_DEVICE_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    double *d() const { return static_cast<double *>(p); }
};

bunmpc::BatchArgs to_args(const bmpc_batch_t &d) {
    bunmpc::BatchArgs a;
    std::memset(&a, 0, sizeof(a));
    a.B = d.B; a.H = d.n_col; a.raw = d.raw; a.cold_start = d.cold_start; a.precision = d.precision;
    a.L0_x = BMPC_L0_X; a.L0_f = BMPC_L0_F;
    a.c.m = d.m; a.c.rho = d.rho; a.c.mu = d.mu; a.c.beta = d.beta;
    a.c.tol = d.tol; a.c.exit_tol = d.exit_tol; a.c.maxit = d.maxit; a.c.num_iters = d.num_iters;
    a.cnt_plan = d.cnt_plan; a.dt = d.dt; a.x_init = d.x_init;
    a.W_X = d.W_X; a.W_X_ter = d.W_X_ter; a.W_F = d.W_F; a.bounds = d.bounds;
    a.X_nom = d.X_nom; a.X_ter = d.X_ter;
    a.sW_X = d.sW_X; a.sW_X_ter = d.sW_X_ter; a.sW_F = d.sW_F; a.sbounds = d.sbounds;
    a.Qx = d.Qx; a.qx = d.qx; a.lbx = d.lbx; a.ubx = d.ubx; a.Qf = d.Qf; a.qf = d.qf;
    a.X = d.X; a.F = d.F; a.P = d.P; a.L_x = d.L_x; a.L_f = d.L_f;
    a.dyn_viol = d.dyn_viol; a.hist = d.hist; a.stats = d.stats; a.trace = d.trace;
    return a;
}

int check_batch(const bmpc_batch_t *d) {
    if (!d) return fail(BMPC_BAD_ARG, "null batch descriptor");
    if (d->B < 0 || d->n_col < 1) return fail(BMPC_BAD_ARG, "B < 0 or n_col < 1");
    if (d->n_eff != 4) return fail(BMPC_BAD_ARG, "only n_eff == 4 is built");
    if (d->n_col + 1 > bunmpc::kMaxKnots)
        return fail(BMPC_BAD_ARG, "n_col + 1 > 256 knots is not supported (one knot per lane, at most four waves per problem)");
    if (d->n_col + 1 > 64 && d->precision != 0) return fail(BMPC_BAD_ARG, "n_col + 1 > 64 knots: fp64 only");
    if (d->num_iters < 0 || d->maxit < 0) return fail(BMPC_BAD_ARG, "negative iteration cap");
    if (d->maxit > bunmpc::kMaxFistaIters) return fail(BMPC_BAD_ARG, "maxit > 4096 is not supported");
    if (d->cold_start < 0 || d->cold_start > 2) return fail(BMPC_BAD_ARG, "cold_start must be 0, 1 or 2");
    if (d->precision != 0 && d->precision != 1) return fail(BMPC_BAD_ARG, "precision must be 0 (fp64) or 1 (fp32)");
    if (d->precision == 1 && d->raw) return fail(BMPC_BAD_ARG, "fp32 arithmetic is built for the harness form only");
    if (!d->cnt_plan || !d->dt || !d->x_init || !d->X || !d->F || !d->P || !d->L_x || !d->L_f)
        return fail(BMPC_BAD_ARG, "missing required array");
    if (d->raw) {
        if (!d->Qx || !d->qx || !d->lbx || !d->ubx || !d->Qf)
            return fail(BMPC_BAD_ARG, "raw form needs Qx, qx, lbx, ubx, Qf");
    } else {
        if (!d->W_X || !d->W_X_ter || !d->W_F || !d->bounds || !d->X_nom || !d->X_ter)
            return fail(BMPC_BAD_ARG, "harness form needs W_X, W_X_ter, W_F, bounds, X_nom, X_ter");
        // batch strides: 0 (one block shared by all problems) or the distance between two problems' blocks, in doubles.  The
        // kernels reach the (at most four) problems of a wave by 32-bit byte offsets from the wave's first problem.
        for (long stride : {d->sW_X, d->sW_X_ter, d->sW_F, d->sbounds})
            if (stride < 0 || stride > (1L << 26)) return fail(BMPC_BAD_ARG, "batch stride of a weight / bounds array is negative or above 2^26 doubles");
    }
    return BMPC_OK;
}

}  // namespace

namespace bunmpc {
int set_error(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace bunmpc

// =============================================================== QuadrupedGait ==
struct bmpc_gait {
    int n_eff;
    double gait_period, step_height;
    std::vector<double> stance_percent, stance_time, swing_time, phase_offset, phi, phase_percent;
    std::vector<int> phase;

    double phi_of(double t, int f) const {  // gait_planner.cpp:41-44
        return std::fmod(t + phase_offset[f] * gait_period, gait_period);
    }
};

// ================================================================== BiConvexMP ==
struct bmpc_biconvex {
    double m;
    int n_col, n_eff;
    // solver parameters (biconvex.hpp:146-160, biconvex.cpp:20-21, fista.hpp:52-60)
    double rho = 1e5, beta = 1.5, mu = 1.0, tol = 1e-5, exit_tol = 1e-3;
    int maxit = 150;
    double L_x = BMPC_L0_X, L_f = BMPC_L0_F;
    // contact arrays (centroidal.hpp:39-52): cnt_arr_/dt_ persist, r_ is cleared by optimize
    std::vector<double> cnt_arr, dt, r;  // cnt_arr [H][E], r [n_set][E][3]
    int n_set = 0;                       // r_.size()
    // problem data (problem.hpp): diagonal Q, q, bounds
    std::vector<double> Qx, qx, lbx, ubx, Qf, qf, lbf, ubf;
    bool qf_nonzero = false;
    // iterates
    std::vector<double> X, F, P;
    std::vector<double> rot;  // set_rotation_matrix_f: stored, unused (as in the reference)
    bool log_statistics = false;
    std::vector<double> hist;
    int last_stats[bunmpc::kStats] = {0, 0, 0, 0, 0, 0};
    DevBuf dbuf, dstats;

    int nx() const { return 9 * (n_col + 1); }
    int nf() const { return 3 * n_eff * n_col; }
};

extern "C" {

int bmpc_abi_version(void) { return 2; }
int bmpc_batch_struct_size(void) { return (int)sizeof(bmpc_batch_t); }
int bmpc_set_three_per_wave(int mode) { return bunmpc::set_three_per_wave(mode); }
int bmpc_set_work_stealing(int on) { return bunmpc::set_work_stealing(on); }
int bmpc_set_steal_grid(int waves) { return bunmpc::set_steal_grid(waves); }
int bmpc_set_two_waves_per_simd(int mode) { return bunmpc::set_two_waves_per_simd(mode); }
int bmpc_biconvex_last_waves_per_simd(void) { return bunmpc::biconvex_last_waves_per_simd(); }
int bmpc_biconvex_last_lanes_per_problem(void) { return bunmpc::biconvex_last_lanes_per_problem(); }
int bmpc_set_latency_mapping_max_batch(int max_batch) { return bunmpc::set_latency_mapping_max_batch(max_batch); }
int bmpc_set_exact_step_decisions(int on) { return bunmpc::set_exact_step_decisions(on); }
int bmpc_biconvex_fp32_scratch_bytes(void) { return bunmpc::biconvex_admm_f32_scratch_bytes(); }
const char *bmpc_last_error(void) { return g_err.c_str(); }

int bmpc_device_count(int *count) {
    if (!count) return fail(BMPC_BAD_ARG, "null count");
    HIP_TRY(hipGetDeviceCount(count));
    return BMPC_OK;
}

int bmpc_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return BMPC_OK;
}

int bmpc_selftest_lanes(void) {
    double in[64], out[12 * 64];
    for (int i = 0; i < 64; ++i) in[i] = (double)(i + 1) + 0.25 * (i % 3);
    DevBuf b;
    HIP_TRY(b.ensure(sizeof(in) + sizeof(out)));
    double *din = b.d(), *dout = b.d() + 64;
    HIP_TRY(hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice));
    HIP_TRY(bunmpc::launch_lane_selftest(din, dout, nullptr));
    HIP_TRY(hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost));
    auto near = [](double a, double b) { return std::fabs(a - b) <= 1e-9 * (1.0 + std::fabs(b)); };
    for (int i = 0; i < 64; ++i) {
        const double prev = i > 0 ? in[i - 1] : 0.0, next = i < 63 ? in[i + 1] : 0.0;
        if (out[i] != prev) return fail(BMPC_DEVICE_ERROR, "from_prev (wave_shr:1) mismatch at lane " + std::to_string(i));
        if (out[64 + i] != next) return fail(BMPC_DEVICE_ERROR, "from_next (wave_shl:1) mismatch at lane " + std::to_string(i));
        for (int w = 0; w < 3; ++w) {
            const int lpp = 16 << w;
            double s = 0;
            for (int j = (i / lpp) * lpp; j < (i / lpp + 1) * lpp; ++j) s += in[j];
            if (!near(out[128 + 64 * w + i], s))
                return fail(BMPC_DEVICE_ERROR, "seg_sum<" + std::to_string(lpp) + "> mismatch at lane " + std::to_string(i));
            if (out[128 + 64 * w + i] != out[128 + 64 * w + (i / lpp) * lpp])
                return fail(BMPC_DEVICE_ERROR, "seg_sum<" + std::to_string(lpp) + "> not segment-uniform");
        }
        if (out[320 + i] != 64.0) return fail(BMPC_DEVICE_ERROR, "ballot popcount mismatch");
    }
    // the 21-lane segments (three problems per wave): sums in every lane of a segment, at the designated lanes, and the spread of
    // a decision taken there
    double s21[3] = {0, 0, 0};
    for (int i = 0; i < 63; ++i) s21[i / 21] += in[i];
    for (int i = 0; i < 63; ++i) {
        if (!near(out[384 + i], s21[i / 21])) return fail(BMPC_DEVICE_ERROR, "seg_sum<21> mismatch at lane " + std::to_string(i));
        if (out[384 + i] != out[384 + (i / 21) * 21]) return fail(BMPC_DEVICE_ERROR, "seg_sum<21> not segment-uniform");
    }
    for (int sgm = 0; sgm < 3; ++sgm) {
        const int d = 16 * (sgm + 1);
        if (!near(out[448 + d], s21[sgm]) || !near(out[512 + d], 2.0 * s21[sgm])) return fail(BMPC_DEVICE_ERROR, "seg_sum2<21> mismatch at its designated lane");
        if (out[448 + d] != out[384 + 21 * sgm]) return fail(BMPC_DEVICE_ERROR, "seg_sum2<21> and seg_sum<21> differ");
    }
    {   // in[i] > 40 at lanes 16 (no), 32 (no: 33.5), 48 (yes: 49) -> only segment 2; lanes 16 and 48 set -> segments 0 and 2
        const unsigned long long seg0 = (1ull << 21) - 1, seg2 = seg0 << 42;
        if (out[576] != (double)seg2) return fail(BMPC_DEVICE_ERROR, "seg_uniform<21> (one segment) mismatch");
        const unsigned long long want = seg0 | seg2;
        if (out[640] != (double)(unsigned)(want >> 32) || out[704] != (double)(unsigned)(want & 0xffffffffu))
            return fail(BMPC_DEVICE_ERROR, "seg_uniform<21> (two segments) mismatch");
    }
    return BMPC_OK;
}

// ------------------------------------------------------------------- gait ----
bmpc_gait_t *bmpc_gait_create(double gait_period, const double *stance_percent,
                              const double *phase_offset, int n_eff, double step_height) {
    if (!stance_percent || !phase_offset || n_eff < 1) { fail(BMPC_BAD_ARG, "bad gait arguments"); return nullptr; }
    auto *g = new bmpc_gait;
    g->n_eff = n_eff; g->gait_period = gait_period; g->step_height = step_height;
    g->stance_percent.assign(stance_percent, stance_percent + n_eff);
    g->phase_offset.assign(phase_offset, phase_offset + n_eff);
    g->stance_time.resize(n_eff); g->swing_time.resize(n_eff);
    g->phi.assign(n_eff, 0.0); g->phase_percent.assign(n_eff, 0.0); g->phase.assign(n_eff, 0);
    for (int i = 0; i < n_eff; ++i) {  // gait_planner.cpp:13-24
        g->stance_time[i] = gait_period * stance_percent[i];
        g->swing_time[i] = gait_period - g->stance_time[i];
    }
    return g;
}
void bmpc_gait_destroy(bmpc_gait_t *g) { delete g; }
int bmpc_gait_n_eff(const bmpc_gait_t *g) { return g ? g->n_eff : 0; }

#define GAIT_FOOT_CHECK()                                                                 \
    if (!g || !out_) return fail(BMPC_BAD_ARG, "null argument");                          \
    if (foot_id < 0 || foot_id >= g->n_eff) return fail(BMPC_BAD_ARG, "foot_id out of range")

int bmpc_gait_get_phi(bmpc_gait_t *g, double t, int foot_id, double *out_) {
    GAIT_FOOT_CHECK();
    *out_ = g->phi_of(t, foot_id);
    return BMPC_OK;
}
int bmpc_gait_get_phi_all(bmpc_gait_t *g, double t, double *phi) {  // gait_planner.cpp:31-39
    if (!g || !phi) return fail(BMPC_BAD_ARG, "null argument");
    for (int i = 0; i < g->n_eff; ++i) g->phi[0] = g->phi_of(t, i);  // sic: only element 0 is written
    std::memcpy(phi, g->phi.data(), sizeof(double) * g->n_eff);
    return BMPC_OK;
}
int bmpc_gait_get_phase(bmpc_gait_t *g, double t, int foot_id, int *out_) {  // gait_planner.cpp:46-58
    GAIT_FOOT_CHECK();
    const double phi = g->phi_of(t, foot_id), st = g->stance_time[foot_id];
    g->phase[foot_id] = (phi <= st || std::fabs(phi - st) < 1e-4) ? 1 : 0;
    *out_ = g->phase[foot_id];
    return BMPC_OK;
}
int bmpc_gait_get_phase_all(bmpc_gait_t *g, double t, int *phase) {  // gait_planner.cpp:60-75 (no slack)
    if (!g || !phase) return fail(BMPC_BAD_ARG, "null argument");
    for (int i = 0; i < g->n_eff; ++i) g->phase[i] = g->phi_of(t, i) <= g->stance_time[i] ? 1 : 0;
    std::memcpy(phase, g->phase.data(), sizeof(int) * g->n_eff);
    return BMPC_OK;
}
int bmpc_gait_get_percent_in_phase(bmpc_gait_t *g, double t, int foot_id, double *out_) {  // :112-128
    GAIT_FOOT_CHECK();
    const double phi = g->phi_of(t, foot_id), st = g->stance_time[foot_id];
    *out_ = phi <= st ? phi / st : (phi - st) / (g->gait_period - st);
    return BMPC_OK;
}
int bmpc_gait_get_percent_in_phase_all(bmpc_gait_t *g, double t, double *pct) {  // :77-96
    if (!g || !pct) return fail(BMPC_BAD_ARG, "null argument");
    for (int i = 0; i < g->n_eff; ++i) g->phi[0] = g->phi_of(t, i);  // get_phi(t) quirk
    for (int i = 0; i < g->n_eff; ++i) {
        const double phi = g->phi[i], st = g->stance_time[i];
        g->phase_percent[i] = phi <= st ? phi / st : (phi - st) / (g->gait_period - st);
    }
    std::memcpy(pct, g->phase_percent.data(), sizeof(double) * g->n_eff);
    return BMPC_OK;
}
int bmpc_gait_get_contact_phase_plan(bmpc_gait_t *g, int rows, double t, double dt, int *plan) {  // :98-104
    if (!g || !plan || rows < 0) return fail(BMPC_BAD_ARG, "bad argument");
    for (int i = 0; i < rows; ++i) {
        int rc = bmpc_gait_get_phase_all(g, t + i * dt, plan + (size_t)i * g->n_eff);
        if (rc) return rc;
    }
    return BMPC_OK;
}
int bmpc_gait_set_step_height(bmpc_gait_t *g, double step_height) {
    if (!g) return fail(BMPC_BAD_ARG, "null handle");
    g->step_height = step_height;
    return BMPC_OK;
}
int bmpc_gait_set_stance_percent(bmpc_gait_t *g, double lf, double lh, double rf, double rh) {  // :130-137
    if (!g || g->n_eff != 4) return fail(BMPC_BAD_ARG, "set_stance_percent needs a 4-foot gait");
    const double v[4] = {lf, lh, rf, rh};
    for (int i = 0; i < 4; ++i) {
        g->stance_percent[i] = v[i];
        g->stance_time[i] = g->gait_period * v[i];
        g->swing_time[i] = g->gait_period - v[i];  // sic (percent, not time); swing_time_ is never read
    }
    return BMPC_OK;
}

// --------------------------------------------------------------- BiconvexMP ----
bmpc_biconvex_t *bmpc_biconvex_create(double m, int n_col, int n_eff) {
    if (n_col < 1 || n_eff < 1) { fail(BMPC_BAD_ARG, "n_col and n_eff must be positive"); return nullptr; }
    auto *h = new bmpc_biconvex;
    h->m = m; h->n_col = n_col; h->n_eff = n_eff;
    h->cnt_arr.assign((size_t)n_col * n_eff, 0.0);
    h->dt.assign(n_col, 0.0);
    h->Qx.assign(h->nx(), 0.0); h->qx.assign(h->nx(), 0.0);
    h->lbx.assign(h->nx(), 0.0); h->ubx.assign(h->nx(), 0.0);   // ProblemData ctor: setZero
    h->Qf.assign(h->nf(), 0.0); h->qf.assign(h->nf(), 0.0);
    h->lbf.assign(h->nf(), 0.0); h->ubf.assign(h->nf(), 0.0);
    h->X.assign(h->nx(), 0.0); h->F.assign(h->nf(), 0.0); h->P.assign(h->nx(), 0.0);
    return h;
}
void bmpc_biconvex_destroy(bmpc_biconvex_t *h) { delete h; }
int bmpc_biconvex_n_col(const bmpc_biconvex_t *h) { return h ? h->n_col : 0; }
int bmpc_biconvex_n_eff(const bmpc_biconvex_t *h) { return h ? h->n_eff : 0; }

#define H_CHECK(...)                                                      \
    if (!h) return fail(BMPC_BAD_ARG, "null handle");                     \
    { const void *ptrs_[] = {__VA_ARGS__};                                \
      for (const void *p_ : ptrs_) if (!p_) return fail(BMPC_BAD_ARG, "null array argument"); }

int bmpc_biconvex_set_contact_plan(bmpc_biconvex_t *h, const double *cnt_plan, double dt) {
    H_CHECK(cnt_plan);
    const int i = h->n_set;  // centroidal.cpp:40-41: r_.push_back, i = r_.size()-1
    if (i >= h->n_col)
        return fail(BMPC_BAD_ARG, "set_contact_plan called more than n_col times since the last optimize");
    h->r.resize((size_t)(i + 1) * h->n_eff * 3);
    for (int j = 0; j < h->n_eff; ++j) {
        h->dt[i] = dt;
        h->cnt_arr[(size_t)i * h->n_eff + j] = cnt_plan[4 * j];
        for (int k = 0; k < 3; ++k) h->r[((size_t)i * h->n_eff + j) * 3 + k] = cnt_plan[4 * j + 1 + k];
    }
    h->n_set = i + 1;
    return BMPC_OK;
}

int bmpc_biconvex_set_rotation_matrix_f(bmpc_biconvex_t *h, const double *R) {
    H_CHECK(R);
    h->rot.insert(h->rot.end(), R, R + 9);
    return BMPC_OK;
}

static int need_plan(const bmpc_biconvex_t *h) {
    if (h->n_set < h->n_col) return fail(BMPC_BAD_ARG, "contact plan incomplete: set_contact_plan must be called n_col times first");
    return BMPC_OK;
}

// dense A_x / b_x exactly as centroidal.cpp:57-84 fills them (debug getters)
static void host_x_mat(const bmpc_biconvex_t *h, const double *X, double *A, double *b) {
    const int H = h->n_col, E = h->n_eff, ncols = 3 * E * H;
    if (A) std::memset(A, 0, sizeof(double) * (size_t)h->nx() * ncols);
    if (b) std::memset(b, 0, sizeof(double) * h->nx());
    for (int t = 0; t < H; ++t) {
        const double dt = h->dt[t];
        if (b) {
            for (int k = 3; k < 9; ++k) b[9 * t + k] = X[9 * (t + 1) + k] - X[9 * t + k];
            b[9 * t + 5] = X[9 * (t + 1) + 5] - X[9 * t + 5] + 9.81 * dt;
        }
        if (!A) continue;
        for (int n = 0; n < E; ++n) {
            const double c = h->cnt_arr[(size_t)t * E + n];
            const double *r = &h->r[((size_t)t * E + n) * 3];
            const int c0 = 3 * E * t + 3 * n;
            auto at = [&](int row, int col) -> double & { return A[(size_t)row * ncols + col]; };
            for (int k = 0; k < 3; ++k) at(9 * t + 3 + k, c0 + k) = c * (dt / h->m);
            at(9 * t + 6, c0 + 1) = c * (X[9 * t + 2] - r[2]) * dt;
            at(9 * t + 6, c0 + 2) = -c * (X[9 * t + 1] - r[1]) * dt;
            at(9 * t + 7, c0 + 0) = -c * (X[9 * t + 2] - r[2]) * dt;
            at(9 * t + 7, c0 + 2) = c * (X[9 * t + 0] - r[0]) * dt;
            at(9 * t + 8, c0 + 0) = c * (X[9 * t + 1] - r[1]) * dt;
            at(9 * t + 8, c0 + 1) = -c * (X[9 * t + 0] - r[0]) * dt;
        }
    }
}

// dense A_f / b_f as centroidal.cpp:6-37,86-127 and centroidal.hpp:22-27 fill them
static void host_f_mat(const bmpc_biconvex_t *h, const double *F, const double *x_init, double *A, double *b) {
    const int H = h->n_col, E = h->n_eff, n = h->nx();
    if (A) std::memset(A, 0, sizeof(double) * (size_t)n * n);
    if (b) std::memset(b, 0, sizeof(double) * n);
    auto at = [&](int row, int col) -> double & { return A[(size_t)row * n + col]; };
    for (int t = 0; t < H; ++t) {
        const double dt = h->dt[t];
        double S[3] = {0, 0, 0}, bl[3] = {0, 0, 0}, ba[3] = {0, 0, 0};
        for (int e = 0; e < E; ++e) {
            const double c = h->cnt_arr[(size_t)t * E + e];
            const double *f = F + 3 * E * t + 3 * e;
            const double *r = &h->r[((size_t)t * E + e) * 3];
            for (int k = 0; k < 3; ++k) { S[k] += c * f[k] * dt; bl[k] += -c * f[k] * dt / h->m; }
            ba[0] += (c * f[1] * r[2] - c * f[2] * r[1]) * dt;
            ba[1] += (c * f[2] * r[0] - c * f[0] * r[2]) * dt;
            ba[2] += (c * f[0] * r[1] - c * f[1] * r[0]) * dt;
        }
        if (A) {
            for (int l = 0; l < 9; ++l) { at(9 * t + l, 9 * t + l) = 1.0; at(9 * t + l, 9 * (t + 1) + l) = -1.0; }
            for (int k = 0; k < 3; ++k) at(9 * t + k, 9 * (t + 1) + 3 + k) = dt;
            at(9 * t + 6, 9 * t + 1) = -S[2]; at(9 * t + 6, 9 * t + 2) = S[1];
            at(9 * t + 7, 9 * t + 0) = S[2];  at(9 * t + 7, 9 * t + 2) = -S[0];
            at(9 * t + 8, 9 * t + 0) = -S[1]; at(9 * t + 8, 9 * t + 1) = S[0];
        }
        if (b) {
            for (int k = 0; k < 3; ++k) { b[9 * t + 3 + k] = bl[k]; b[9 * t + 6 + k] = ba[k]; }
            b[9 * t + 5] += 9.81 * dt;
        }
    }
    for (int k = 0; k < 9; ++k) {
        if (A) at(9 * H + k, k) = 1.0;
        if (b) b[9 * H + k] = x_init[k];
    }
}

int bmpc_biconvex_return_A_x(bmpc_biconvex_t *h, const double *X, double *A_x) {
    H_CHECK(X, A_x);
    if (int rc = need_plan(h)) return rc;
    host_x_mat(h, X, A_x, nullptr);
    return BMPC_OK;
}
int bmpc_biconvex_return_b_x(bmpc_biconvex_t *h, const double *X, double *b_x) {
    H_CHECK(X, b_x);
    if (int rc = need_plan(h)) return rc;
    host_x_mat(h, X, nullptr, b_x);
    return BMPC_OK;
}
int bmpc_biconvex_return_A_f(bmpc_biconvex_t *h, const double *F, const double *x_init, double *A_f) {
    H_CHECK(F, x_init, A_f);
    if (int rc = need_plan(h)) return rc;
    host_f_mat(h, F, x_init, A_f, nullptr);
    return BMPC_OK;
}
int bmpc_biconvex_return_b_f(bmpc_biconvex_t *h, const double *F, const double *x_init, double *b_f) {
    H_CHECK(F, x_init, b_f);
    if (int rc = need_plan(h)) return rc;
    host_f_mat(h, F, x_init, nullptr, b_f);
    return BMPC_OK;
}

int bmpc_biconvex_set_cost_x(bmpc_biconvex_t *h, const double *Q, const double *q) {
    H_CHECK(Q, q);
    h->Qx.assign(Q, Q + h->nx()); h->qx.assign(q, q + h->nx());
    return BMPC_OK;
}
int bmpc_biconvex_set_cost_f(bmpc_biconvex_t *h, const double *Q, const double *q) {
    H_CHECK(Q, q);
    h->Qf.assign(Q, Q + h->nf()); h->qf.assign(q, q + h->nf());
    h->qf_nonzero = false;
    for (double v : h->qf) if (v != 0.0) h->qf_nonzero = true;
    return BMPC_OK;
}
int bmpc_biconvex_create_cost_X(bmpc_biconvex_t *h, const double *W_X, const double *W_X_ter,
                                const double *X_ter, const double *X_nom) {  // biconvex.cpp:57-72
    H_CHECK(W_X, W_X_ter, X_ter, X_nom);
    const int nv = h->nx();
    for (int i = 0; i < nv - 9; ++i) { h->Qx[i] = W_X[i]; h->qx[i] = -2 * (X_nom[i] * W_X[i]); }
    for (int i = nv - 9; i < nv; ++i) {
        h->Qx[i] = W_X_ter[i - nv + 9];
        h->qx[i] = -2 * (X_ter[i - nv + 9] * W_X_ter[i - nv + 9]);
    }
    return BMPC_OK;
}
int bmpc_biconvex_create_cost_F(bmpc_biconvex_t *h, const double *W_F) {  // biconvex.cpp:74-78
    H_CHECK(W_F);
    h->Qf.assign(W_F, W_F + h->nf());
    return BMPC_OK;
}
int bmpc_biconvex_set_bounds_x(bmpc_biconvex_t *h, const double *lb, const double *ub) {
    H_CHECK(lb, ub);
    h->lbx.assign(lb, lb + h->nx()); h->ubx.assign(ub, ub + h->nx());
    return BMPC_OK;
}
int bmpc_biconvex_set_bounds_f(bmpc_biconvex_t *h, const double *lb, const double *ub) {
    H_CHECK(lb, ub);
    h->lbf.assign(lb, lb + h->nf()); h->ubf.assign(ub, ub + h->nf());  // unused under the SoC projection
    return BMPC_OK;
}
int bmpc_biconvex_create_bound_constraints(bmpc_biconvex_t *h, const double *b, int rows, int cols,
                                           double fx_max, double fy_max, double fz_max) {  // biconvex.cpp:27-55
    H_CHECK(b);
    if (cols != 6) {
        std::cout << "bound constraints wrong size. Expected 6 ..." << std::endl;  // biconvex.cpp:33-35
        return fail(BMPC_BAD_ARG, "bound matrix must have 6 columns");
    }
    if (rows < h->n_col) return fail(BMPC_BAD_ARG, "bound matrix needs n_col rows");
    if (int rc = need_plan(h)) return rc;
    const double inf = std::numeric_limits<double>::infinity();
    const int H = h->n_col, E = h->n_eff;
    for (int i = 0; i < h->nx(); ++i) { h->lbx[i] = -inf; h->ubx[i] = inf; }
    for (int i = 0; i < H; ++i) {
        double csum = 0;
        for (int j = 0; j < E; ++j) {
            h->lbf[3 * E * i + 3 * j] = -fx_max; h->lbf[3 * E * i + 3 * j + 1] = -fy_max; h->lbf[3 * E * i + 3 * j + 2] = 0;
            h->ubf[3 * E * i + 3 * j] = fx_max;  h->ubf[3 * E * i + 3 * j + 1] = fy_max;  h->ubf[3 * E * i + 3 * j + 2] = fz_max;
            csum += h->cnt_arr[(size_t)i * E + j];
        }
        if (csum > 0) {
            for (int k = 0; k < 3; ++k) {
                double mx = -inf, mn = inf;
                for (int j = 0; j < E; ++j) {
                    const double v = h->r[((size_t)i * E + j) * 3 + k];
                    mx = v > mx ? v : mx; mn = v < mn ? v : mn;
                }
                h->lbx[9 * i + k] = mx + b[6 * i + k];
                h->ubx[9 * i + k] = mn + b[6 * i + 3 + k];
            }
        }
    }
    return BMPC_OK;
}
int bmpc_biconvex_set_rho(bmpc_biconvex_t *h, double rho) { H_CHECK(h); h->rho = rho; return BMPC_OK; }
int bmpc_biconvex_set_friction_coefficient(bmpc_biconvex_t *h, double mu) { H_CHECK(h); h->mu = mu; return BMPC_OK; }
int bmpc_biconvex_set_robot_mass(bmpc_biconvex_t *h, double m) { H_CHECK(h); h->m = m; return BMPC_OK; }

int bmpc_biconvex_return_opt_x(bmpc_biconvex_t *h, double *X) { H_CHECK(X); std::memcpy(X, h->X.data(), sizeof(double) * h->nx()); return BMPC_OK; }
int bmpc_biconvex_return_opt_f(bmpc_biconvex_t *h, double *F) { H_CHECK(F); std::memcpy(F, h->F.data(), sizeof(double) * h->nf()); return BMPC_OK; }
int bmpc_biconvex_return_opt_p(bmpc_biconvex_t *h, double *P) { H_CHECK(P); std::memcpy(P, h->P.data(), sizeof(double) * h->nx()); return BMPC_OK; }
int bmpc_biconvex_return_opt_com(bmpc_biconvex_t *h, double *com) {  // biconvex.cpp:122-130
    H_CHECK(com);
    for (int i = 0; i <= h->n_col; ++i) for (int k = 0; k < 3; ++k) com[3 * i + k] = h->X[9 * i + k];
    return BMPC_OK;
}
int bmpc_biconvex_return_opt_mom(bmpc_biconvex_t *h, double *mom) {  // biconvex.cpp:132-142
    H_CHECK(mom);
    for (int i = 0; i <= h->n_col; ++i) {
        for (int k = 0; k < 3; ++k) mom[6 * i + k] = h->m * h->X[9 * i + 3 + k];
        for (int k = 0; k < 3; ++k) mom[6 * i + 3 + k] = h->X[9 * i + 6 + k];
    }
    return BMPC_OK;
}
int bmpc_biconvex_set_warm_start_vars(bmpc_biconvex_t *h, const double *X, const double *F, const double *P) {
    H_CHECK(X, F, P);
    h->X.assign(X, X + h->nx()); h->F.assign(F, F + h->nf()); h->P.assign(P, P + h->nx());
    return BMPC_OK;
}
int bmpc_biconvex_dyn_viol_hist_size(const bmpc_biconvex_t *h) { return h ? (int)h->hist.size() : 0; }
int bmpc_biconvex_return_dyn_viol_hist(const bmpc_biconvex_t *h, double *hist) {
    H_CHECK(hist);
    std::memcpy(hist, h->hist.data(), sizeof(double) * h->hist.size());
    return BMPC_OK;
}
int bmpc_biconvex_collect_statistics(bmpc_biconvex_t *h) { H_CHECK(h); h->log_statistics = true; return BMPC_OK; }
int bmpc_biconvex_get_step_constants(const bmpc_biconvex_t *h, double *L_x, double *L_f) {
    H_CHECK(L_x, L_f);
    *L_x = h->L_x; *L_f = h->L_f;
    return BMPC_OK;
}
int bmpc_biconvex_set_step_constants(bmpc_biconvex_t *h, double L_x, double L_f) {
    H_CHECK(h);
    h->L_x = L_x; h->L_f = L_f;
    return BMPC_OK;
}
int bmpc_biconvex_last_stats(const bmpc_biconvex_t *h, int *stats6) {
    H_CHECK(stats6);
    std::memcpy(stats6, h->last_stats, sizeof(h->last_stats));
    return BMPC_OK;
}

// BiConvexMP::optimize (biconvex.cpp:80-120) as one B = 1 launch of the batched kernel.
int bmpc_biconvex_optimize(bmpc_biconvex_t *h, const double *x_init, int num_iters) {
    H_CHECK(x_init);
    if (int rc = need_plan(h)) return rc;
    if (num_iters < 0) return fail(BMPC_BAD_ARG, "num_iters < 0");
    const int H = h->n_col, E = h->n_eff, nx = h->nx(), nf = h->nf();
    // pack host -> one staging vector -> device (single H2D copy)
    std::vector<double> cnt((size_t)H * E * 4);
    for (int t = 0; t < H; ++t)
        for (int n = 0; n < E; ++n) {
            cnt[((size_t)t * E + n) * 4] = h->cnt_arr[(size_t)t * E + n];
            for (int k = 0; k < 3; ++k) cnt[((size_t)t * E + n) * 4 + 1 + k] = h->r[((size_t)t * E + n) * 3 + k];
        }
    const int nh = num_iters > 0 ? num_iters : 1;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += n; return o; };
    const size_t o_cnt = take(cnt.size()), o_dt = take(H), o_xi = take(9), o_Qx = take(nx), o_qx = take(nx),
                 o_lb = take(nx), o_ub = take(nx), o_Qf = take(nf), o_qf = take(nf), o_X = take(nx),
                 o_F = take(nf), o_P = take(nx), o_L = take(2), o_viol = take(1), o_hist = take(nh);
    std::vector<double> stage(off, 0.0);
    auto put = [&](size_t o, const double *src, size_t n) { std::memcpy(stage.data() + o, src, sizeof(double) * n); };
    put(o_cnt, cnt.data(), cnt.size()); put(o_dt, h->dt.data(), H); put(o_xi, x_init, 9);
    put(o_Qx, h->Qx.data(), nx); put(o_qx, h->qx.data(), nx); put(o_lb, h->lbx.data(), nx); put(o_ub, h->ubx.data(), nx);
    put(o_Qf, h->Qf.data(), nf); put(o_qf, h->qf.data(), nf);
    put(o_X, h->X.data(), nx); put(o_F, h->F.data(), nf); put(o_P, h->P.data(), nx);
    stage[o_L] = h->L_x; stage[o_L + 1] = h->L_f;
    HIP_TRY(h->dbuf.ensure(sizeof(double) * off));
    HIP_TRY(h->dstats.ensure(sizeof(int) * bunmpc::kStats));
    double *d = h->dbuf.d();
    HIP_TRY(hipMemcpy(d, stage.data(), sizeof(double) * off, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->dstats.p, 0, sizeof(int) * bunmpc::kStats));

    bmpc_batch_t b;
    std::memset(&b, 0, sizeof(b));
    b.B = 1; b.n_col = H; b.n_eff = E; b.raw = 1; b.num_iters = num_iters; b.maxit = h->maxit;
    b.m = h->m; b.rho = h->rho; b.mu = h->mu; b.beta = h->beta; b.tol = h->tol; b.exit_tol = h->exit_tol;
    b.cnt_plan = d + o_cnt; b.dt = d + o_dt; b.x_init = d + o_xi;
    b.Qx = d + o_Qx; b.qx = d + o_qx; b.lbx = d + o_lb; b.ubx = d + o_ub; b.Qf = d + o_Qf;
    b.qf = h->qf_nonzero ? d + o_qf : nullptr;
    b.X = d + o_X; b.F = d + o_F; b.P = d + o_P; b.L_x = d + o_L; b.L_f = d + o_L + 1;
    b.dyn_viol = d + o_viol; b.hist = d + o_hist; b.stats = static_cast<int *>(h->dstats.p);
    if (int rc = check_batch(&b)) return rc;
    HIP_TRY(bunmpc::launch_biconvex_admm(to_args(b), E, nullptr));
    HIP_TRY(hipMemcpy(stage.data() + o_X, d + o_X, sizeof(double) * (off - o_X), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->last_stats, h->dstats.p, sizeof(int) * bunmpc::kStats, hipMemcpyDeviceToHost));
    std::memcpy(h->X.data(), stage.data() + o_X, sizeof(double) * nx);
    std::memcpy(h->F.data(), stage.data() + o_F, sizeof(double) * nf);
    std::memcpy(h->P.data(), stage.data() + o_P, sizeof(double) * nx);
    h->L_x = stage[o_L]; h->L_f = stage[o_L + 1];
    if (h->log_statistics)
        for (int i = 0; i < h->last_stats[0]; ++i) h->hist.push_back(stage[o_hist + i]);
    h->n_set = 0;  // centroidal_dynamics.r_.clear()  (biconvex.cpp:117)
    h->r.clear();
    if (h->last_stats[5] == 2) {
        std::cout << "ERROR: solver diverged, Dyn violation is NaN" << std::endl;  // biconvex.cpp:107
        return fail(BMPC_DIVERGED, "dynamics violation is NaN");
    }
    return BMPC_OK;
}

// -------------------------------------------------------------------- batch ----
void bmpc_batch_defaults(bmpc_batch_t *d) {
    if (!d) return;
    std::memset(d, 0, sizeof(*d));
    d->n_eff = 4; d->num_iters = 10; d->maxit = 150;
    d->rho = 1e5; d->mu = 1.0; d->beta = 1.5; d->tol = 1e-5; d->exit_tol = 1e-3;
}

int bmpc_biconvex_solve_batch_device(const bmpc_batch_t *d, void *hip_stream) {
    if (int rc = check_batch(d)) return rc;
    HIP_TRY(bunmpc::launch_biconvex_admm(to_args(*d), d->n_eff, static_cast<hipStream_t>(hip_stream)));
    return BMPC_OK;
}

int bmpc_biconvex_solve_batch_host(const bmpc_batch_t *d) {
    if (int rc = check_batch(d)) return rc;
    const size_t B = (size_t)d->B, H = (size_t)d->n_col, E = (size_t)d->n_eff;
    const size_t nx = 9 * (H + 1), nf = 3 * E * H;
    if (B == 0) return BMPC_OK;
    bmpc_batch_t b = *d;
    struct In { const double **slot; size_t n; };
    struct Out { double **slot; double *host; size_t n; };
    auto rows = [&](long stride) { return stride == 0 ? (size_t)1 : B; };
    std::vector<In> ins = {{&b.cnt_plan, B * H * E * 4}, {&b.dt, B * H}, {&b.x_init, B * 9}};
    if (d->raw) {
        ins.push_back({&b.Qx, B * nx}); ins.push_back({&b.qx, B * nx});
        ins.push_back({&b.lbx, B * nx}); ins.push_back({&b.ubx, B * nx});
        ins.push_back({&b.Qf, B * nf});
        if (d->qf) ins.push_back({&b.qf, B * nf});
    } else {
        ins.push_back({&b.W_X, (rows(d->sW_X) - 1) * (size_t)d->sW_X + 9 * H});
        ins.push_back({&b.W_X_ter, (rows(d->sW_X_ter) - 1) * (size_t)d->sW_X_ter + 9});
        ins.push_back({&b.W_F, (rows(d->sW_F) - 1) * (size_t)d->sW_F + nf});
        ins.push_back({&b.bounds, (rows(d->sbounds) - 1) * (size_t)d->sbounds + 6 * H});
        ins.push_back({&b.X_nom, B * 9 * H}); ins.push_back({&b.X_ter, B * 9});
    }
    std::vector<Out> outs = {{&b.X, d->X, B * nx}, {&b.F, d->F, B * nf}, {&b.P, d->P, B * nx},
                             {&b.L_x, d->L_x, B}, {&b.L_f, d->L_f, B}};
    if (d->dyn_viol) outs.push_back({&b.dyn_viol, d->dyn_viol, B});
    if (d->hist) outs.push_back({&b.hist, d->hist, B * (size_t)(d->num_iters > 0 ? d->num_iters : 1)});
    size_t total = 0;
    for (auto &i : ins) total += i.n;
    for (auto &o : outs) total += o.n;
    DevBuf buf, sbuf, tbuf;
    const size_t ntrace = B * (size_t)(d->num_iters > 0 ? d->num_iters : 1) * 4;
    HIP_TRY(buf.ensure(sizeof(double) * total));
    double *p = buf.d();
    for (auto &i : ins) {
        HIP_TRY(hipMemcpy(p, *i.slot, sizeof(double) * i.n, hipMemcpyHostToDevice));
        *i.slot = p; p += i.n;
    }
    for (auto &o : outs) {
        HIP_TRY(hipMemcpy(p, o.host, sizeof(double) * o.n, hipMemcpyHostToDevice));
        *o.slot = p; p += o.n;
    }
    if (d->stats) {
        HIP_TRY(sbuf.ensure(sizeof(int) * bunmpc::kStats * B));
        HIP_TRY(hipMemset(sbuf.p, 0, sizeof(int) * bunmpc::kStats * B));
        b.stats = static_cast<int *>(sbuf.p);
    }
    if (d->trace) {      // rows of ADMM iterations that do not run keep the caller's values
        HIP_TRY(tbuf.ensure(sizeof(int) * ntrace));
        HIP_TRY(hipMemcpy(tbuf.p, d->trace, sizeof(int) * ntrace, hipMemcpyHostToDevice));
        b.trace = static_cast<int *>(tbuf.p);
    }
    HIP_TRY(bunmpc::launch_biconvex_admm(to_args(b), d->n_eff, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    for (auto &o : outs) HIP_TRY(hipMemcpy(o.host, *o.slot, sizeof(double) * o.n, hipMemcpyDeviceToHost));
    if (d->stats) HIP_TRY(hipMemcpy(d->stats, sbuf.p, sizeof(int) * bunmpc::kStats * B, hipMemcpyDeviceToHost));
    if (d->trace) HIP_TRY(hipMemcpy(d->trace, tbuf.p, sizeof(int) * ntrace, hipMemcpyDeviceToHost));
    return BMPC_OK;
}

const char *bmpc_biconvex_kernel_name(int n_col, int raw) { return bunmpc::biconvex_kernel_name(n_col, raw); }
const char *bmpc_biconvex_last_kernel_name(void) { return bunmpc::biconvex_last_kernel_name(); }

}  // extern "C"

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
//     workgroups, B*LPP/64 of them -- no LDS, no barriers, no inter-workgroup traffic;
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
//   * fp64 throughout (MFMA has no advantage over VALU for fp64 on gfx950 and the
//     blocks are 6x12 / 9x9 sparse), iterate state in VGPRs; X / F round-trip
//     through their (L2-resident) output buffers at phase boundaries to keep the
//     register footprint of each FISTA loop small.
#include "biconvex_kernels.h"

namespace bunmpc {
namespace {

#define UNROLL _Pragma("unroll")

// DPP controls (LLVM SIDefines.h DppCtrl)
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_WAVE_SHL1 = 0x130;    // lane i <- lane i+1
constexpr int DPP_WAVE_SHR1 = 0x138;    // lane i <- lane i-1

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// value held by the previous / next knot's lane (0 at the wave ends)
__device__ __forceinline__ double from_prev(double v) { return dpp_mov<DPP_WAVE_SHR1>(v); }
__device__ __forceinline__ double from_next(double v) { return dpp_mov<DPP_WAVE_SHL1>(v); }

__device__ __forceinline__ double swap16_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double swap32_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// Sum over the LPP-lane segment, result in every lane of it (fixed butterfly order,
// so all lanes hold the same bits).
template <int LPP>
__device__ __forceinline__ double seg_sum(double v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    if (LPP >= 32) v = swap16_sum(v);
    if (LPP >= 64) v = swap32_sum(v);
    return v;
}

__device__ __forceinline__ double ldz(const double *p, long i, bool ok) { return ok ? p[i] : 0.0; }

constexpr double kGravity = 9.81;  // centroidal.cpp:63

// ------------------------------------------------------------------------------
template <int LPP, int E, bool RAW>
__global__ __launch_bounds__(64) void biconvex_admm_kernel(const BatchArgs a) {
    constexpr int NF = 3 * E;           // force variables per knot
    constexpr int NB = RAW ? 9 : 3;     // bounded components per knot
    const int lane = threadIdx.x & 63;
    const int t = lane % LPP;           // knot owned by this lane
    const int seg = lane / LPP;
    const int H = a.H;
    const long prob = (long)blockIdx.x * (64 / LPP) + seg;
    const bool pvalid = prob < a.B;
    const bool kvalid = pvalid && t <= H;  // owns knot t (X block t)
    const bool rvalid = pvalid && t < H;   // owns dynamics row-block t and force block t
    const bool l0 = pvalid && t == 0;      // also owns the x_init rows 9H..9H+8
    const long nx = 9L * (H + 1), nf = (long)NF * H;
    const long pb = pvalid ? prob : 0;

    const double m = a.c.m, rho = a.c.rho, mu = a.c.mu, beta = a.c.beta;
    const double tol = a.c.tol, exit_tol = a.c.exit_tol;
    const int maxit = a.c.maxit;
    const double rho2 = 2.0 * rho;

    double *Xg = a.X + pb * nx + 9L * t;
    double *Fg = a.F + pb * nf + (long)NF * t;

    // ---- per-knot constants that live through the whole solve
    const double dt = ldz(a.dt, pb * H + t, rvalid);
    const double dtp = from_prev(dt);  // dt of knot t-1 (0 for t == 0: previous lane is a dead/terminal lane)
    const bool cold = a.cold_start != 0;
    double P[9], PI[9], xin[9];
    UNROLL for (int l = 0; l < 9; ++l) {
        P[l] = cold ? 0.0 : ldz(a.P, pb * nx + 9L * t + l, rvalid);
        PI[l] = cold ? 0.0 : ldz(a.P, pb * nx + 9L * H + l, l0);
        xin[l] = ldz(a.x_init, pb * 9 + l, l0);
    }
    double L_x = cold ? a.L0_x : (pvalid ? a.L_x[pb] : 1.0);
    double L_f = cold ? a.L0_f : (pvalid ? a.L_f[pb] : 1.0);
    if (cold) {  // KinoDynMP::set_warm_starts (kino_dyn.cpp:83-99): X = tile(x_init), F = 0 (P = 0 above)
        if (kvalid) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = a.x_init[pb * 9 + l]; }
        if (rvalid) { UNROLL for (int j = 0; j < NF; ++j) Fg[j] = 0.0; }
    }
    bool alive = pvalid;
    int n_admm = 0, it_f = 0, it_x = 0, bt_f = 0, bt_x = 0, status = 0;
    double last_viol = 0.0;

    for (int it = 0; it < a.c.num_iters; ++it) {
        if (!__any(alive)) break;
        // contact data of this knot: flags c_n, positions r_n  (centroidal.cpp:39-49)
        double c[E], r[E][3];
        UNROLL for (int n = 0; n < E; ++n) {
            const long o = ((pb * H + t) * E + n) * 4;
            c[n] = ldz(a.cnt_plan, o, rvalid);
            UNROLL for (int k = 0; k < 3; ++k) r[n][k] = ldz(a.cnt_plan, o + 1 + k, rvalid);
        }

        // =================================================================== F step
        {
            double X[9];
            UNROLL for (int l = 0; l < 9; ++l) X[l] = kvalid ? Xg[l] : 0.0;
            // bPk rows 9t+3..8 = -b_x + P, b_x = X_{t+1} - X_t (+g dt)   (centroidal.cpp:60-65)
            double bpk[6];
            UNROLL for (int k = 0; k < 6; ++k) {
                const double xn = from_next(X[3 + k]);
                double bx = xn - X[3 + k];
                if (k == 2) bx += kGravity * dt;
                bpk[k] = rvalid ? (-bx + P[3 + k]) : 0.0;
            }
            // A_x entries of this knot (centroidal.cpp:67-81)
            double an[E], sp[E][3];
            UNROLL for (int n = 0; n < E; ++n) {
                an[n] = c[n] * (dt / m);
                UNROLL for (int k = 0; k < 3; ++k) sp[n][k] = c[n] * (X[k] - r[n][k]) * dt;
            }
            double wf2[NF], qf[NF];
            UNROLL for (int j = 0; j < NF; ++j) {
                if (RAW) {
                    wf2[j] = 2.0 * ldz(a.Qf, pb * nf + (long)NF * t + j, rvalid);
                    qf[j] = a.qf ? ldz(a.qf, pb * nf + (long)NF * t + j, rvalid) : 0.0;
                } else {
                    wf2[j] = 2.0 * ldz(a.W_F, pb * a.sW_F + (long)NF * t + j, rvalid);
                    qf[j] = 0.0;
                }
            }
            // u = A v + bPk on rows 9t+3..8
            auto applyA = [&](const double (&v)[NF], double (&u)[6]) {
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
                UNROLL for (int n = 0; n < E; ++n) {
                    const double vx = v[3 * n], vy = v[3 * n + 1], vz = v[3 * n + 2];
                    s0 += an[n] * vx; s1 += an[n] * vy; s2 += an[n] * vz;
                    s3 += sp[n][2] * vy - sp[n][1] * vz;
                    s4 += sp[n][0] * vz - sp[n][2] * vx;
                    s5 += sp[n][1] * vx - sp[n][0] * vy;
                }
                u[0] = s0 + bpk[0]; u[1] = s1 + bpk[1]; u[2] = s2 + bpk[2];
                u[3] = s3 + bpk[3]; u[4] = s4 + bpk[4]; u[5] = s5 + bpk[5];
            };

            double x[NF], y[NF], g[NF], y1[NF];
            double ry[6], rx[6], r1[6];
            UNROLL for (int j = 0; j < NF; ++j) { x[j] = rvalid ? Fg[j] : 0.0; y[j] = x[j]; }
            applyA(y, ry);
            UNROLL for (int k = 0; k < 6; ++k) rx[k] = ry[k];
            const double imu = 1.0 / (mu * mu + 1.0);
            double tk = 1.0;
            bool act = alive;
            for (int i = 0; i < maxit; ++i) {
                if (!__any(act)) break;
                // gradient 2 Q y + q + 2 rho A^T (A y + bPk)          (problem.cpp:36-38,54-56)
                UNROLL for (int n = 0; n < E; ++n) {
                    const double zx = an[n] * ry[0] - sp[n][2] * ry[4] + sp[n][1] * ry[5];
                    const double zy = an[n] * ry[1] + sp[n][2] * ry[3] - sp[n][0] * ry[5];
                    const double zz = an[n] * ry[2] - sp[n][1] * ry[3] + sp[n][0] * ry[4];
                    g[3 * n] = wf2[3 * n] * y[3 * n] + qf[3 * n] + rho2 * zx;
                    g[3 * n + 1] = wf2[3 * n + 1] * y[3 * n + 1] + qf[3 * n + 1] + rho2 * zy;
                    g[3 * n + 2] = wf2[3 * n + 2] * y[3 * n + 2] + qf[3 * n + 2] + rho2 * zz;
                }
                double Gn;
                bool pend = act;
                do {  // backtracking (fista.cpp:8-26); segments that accepted recompute the same values
                    const double invL = 1.0 / L_f;
                    // "SoC" projection exactly as fista.cpp:52-70 writes it
                    UNROLL for (int n = 0; n < E; ++n) {
                        double fx = y[3 * n] - g[3 * n] * invL;
                        double fy = y[3 * n + 1] - g[3 * n + 1] * invL;
                        double fz = y[3 * n + 2] - g[3 * n + 2] * invL;
                        const double s = fx * fx + fy * fy;
                        const double z = fz;
                        const bool zero = (s * mu < -z) || (z < 0);
                        const bool cone = !zero && (s > mu * z);
                        const double k = ((mu * mu) * s + (mu * z)) / (((mu * mu) + 1.0) * s);
                        const double zc = (mu * s + z) * imu;
                        fx = zero ? 0.0 : (cone ? fx * k : fx);
                        fy = zero ? 0.0 : (cone ? fy * k : fy);
                        fz = zero ? 0.0 : (cone ? zc : fz);
                        y1[3 * n] = fx; y1[3 * n + 1] = fy; y1[3 * n + 2] = fz;
                    }
                    applyA(y1, r1);
                    double g2 = 0, gd = 0, od = 0, n1 = 0, n0 = 0;
                    UNROLL for (int j = 0; j < NF; ++j) {
                        const double d = y1[j] - y[j];
                        g2 += d * d;
                        gd += g[j] * d;
                        od += ((y1[j] + y[j]) * (0.5 * wf2[j]) + qf[j]) * d;
                    }
                    UNROLL for (int k = 0; k < 6; ++k) { n1 += r1[k] * r1[k]; n0 += ry[k] * ry[k]; }
                    od += rho * (n1 - n0);
                    g2 = seg_sum<LPP>(g2);
                    gd = seg_sum<LPP>(gd);
                    od = seg_sum<LPP>(od);
                    Gn = sqrt(g2);
                    const bool bt = pend && (od > gd + (L_f * 0.5) * (Gn * Gn));
                    if (bt) { L_f *= beta; ++bt_f; }
                    pend = bt;
                } while (__any(pend));
                // momentum (fista.cpp:33-47); A-images follow by linearity
                const double tk1 = 1.0 + sqrt(1.0 + 4.0 * tk * tk) * 0.5;  // sic
                const double cm = (tk - 1.0) / tk1;
                const bool done = Gn < tol;
                const bool adv = act && !done;
                UNROLL for (int j = 0; j < NF; ++j) {
                    const double yn = y1[j] + cm * (y1[j] - x[j]);
                    x[j] = act ? y1[j] : x[j];
                    y[j] = adv ? yn : y[j];
                }
                UNROLL for (int k = 0; k < 6; ++k) {
                    const double rn = r1[k] + cm * (r1[k] - rx[k]);
                    rx[k] = act ? r1[k] : rx[k];
                    ry[k] = adv ? rn : ry[k];
                }
                tk = adv ? tk1 : tk;
                it_f += act ? 1 : 0;
                act = adv;
            }
            if (rvalid) { UNROLL for (int j = 0; j < NF; ++j) Fg[j] = x[j]; }
        }

        // =================================================================== X step
        {
            // A_f / b_f entries of this knot from the new forces (centroidal.cpp:86-127)
            double SX = 0, SY = 0, SZ = 0, bf[9];
            {
                double b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0;
                UNROLL for (int n = 0; n < E; ++n) {
                    const double fx = rvalid ? Fg[3 * n] : 0.0, fy = rvalid ? Fg[3 * n + 1] : 0.0,
                                 fz = rvalid ? Fg[3 * n + 2] : 0.0;
                    SX += c[n] * fx * dt; SY += c[n] * fy * dt; SZ += c[n] * fz * dt;
                    b3 += -c[n] * fx * dt / m; b4 += -c[n] * fy * dt / m; b5 += -c[n] * fz * dt / m;
                    b6 += (c[n] * fy * r[n][2] - c[n] * fz * r[n][1]) * dt;
                    b7 += (c[n] * fz * r[n][0] - c[n] * fx * r[n][2]) * dt;
                    b8 += (c[n] * fx * r[n][1] - c[n] * fy * r[n][0]) * dt;
                }
                bf[0] = 0; bf[1] = 0; bf[2] = 0;
                bf[3] = b3; bf[4] = b4; bf[5] = b5 + kGravity * dt;
                bf[6] = b6; bf[7] = b7; bf[8] = b8;
            }
            double bpk[9], bpi[9];
            UNROLL for (int l = 0; l < 9; ++l) {
                bpk[l] = rvalid ? (-bf[l] + P[l]) : 0.0;
                bpi[l] = l0 ? (-xin[l] + PI[l]) : 0.0;
            }
            // cost and bounds of this knot
            double q2[9], q[9], lb[NB], ub[NB];
            if (RAW) {
                UNROLL for (int l = 0; l < 9; ++l) {
                    q2[l] = 2.0 * ldz(a.Qx, pb * nx + 9L * t + l, kvalid);
                    q[l] = ldz(a.qx, pb * nx + 9L * t + l, kvalid);
                }
                UNROLL for (int l = 0; l < NB; ++l) {
                    lb[l] = kvalid ? a.lbx[pb * nx + 9L * t + l] : -INFINITY;
                    ub[l] = kvalid ? a.ubx[pb * nx + 9L * t + l] : INFINITY;
                }
            } else {
                // create_cost_X (biconvex.cpp:57-72)
                UNROLL for (int l = 0; l < 9; ++l) {
                    const double w = rvalid ? a.W_X[pb * a.sW_X + 9L * t + l]
                                            : (kvalid ? a.W_X_ter[pb * a.sW_X_ter + l] : 0.0);
                    const double xr = rvalid ? a.X_nom[pb * 9L * H + 9L * t + l]
                                             : (kvalid ? a.X_ter[pb * 9 + l] : 0.0);
                    q2[l] = 2.0 * w;
                    q[l] = -2.0 * (xr * w);
                }
                // create_bound_constraints (biconvex.cpp:27-55): CoM box around the feet
                double csum = 0;
                UNROLL for (int n = 0; n < E; ++n) csum += c[n];
                const bool bounded = rvalid && csum > 0;
                UNROLL for (int k = 0; k < 3; ++k) {
                    double mx = r[0][k], mn = r[0][k];
                    UNROLL for (int n = 1; n < E; ++n) { mx = fmax(mx, r[n][k]); mn = fmin(mn, r[n][k]); }
                    const double blo = bounded ? a.bounds[pb * a.sbounds + 6L * t + k] : 0.0;
                    const double bhi = bounded ? a.bounds[pb * a.sbounds + 6L * t + 3 + k] : 0.0;
                    lb[k] = bounded ? mx + blo : -INFINITY;
                    ub[k] = bounded ? mn + bhi : INFINITY;
                }
            }
            // u = A_f v + bPk on row-block t; vn = v of knot t+1
            auto applyA = [&](const double (&v)[9], double (&u)[9]) {
                double vn[9];
                UNROLL for (int l = 0; l < 9; ++l) vn[l] = from_next(v[l]);
                double w[9];
                UNROLL for (int l = 0; l < 9; ++l) w[l] = v[l] - vn[l];
                UNROLL for (int k = 0; k < 3; ++k) w[k] += dt * vn[3 + k];
                w[6] += SY * v[2] - SZ * v[1];
                w[7] += SZ * v[0] - SX * v[2];
                w[8] += SX * v[1] - SY * v[0];
                UNROLL for (int l = 0; l < 9; ++l) u[l] = rvalid ? (w[l] + bpk[l]) : 0.0;
            };

            double x[9], y[9], g[9], y1[9], ry[9], rx[9], r1[9];
            UNROLL for (int l = 0; l < 9; ++l) { x[l] = kvalid ? Xg[l] : 0.0; y[l] = x[l]; }
            applyA(y, ry);
            UNROLL for (int l = 0; l < 9; ++l) rx[l] = ry[l];
            double tk = 1.0;
            bool act = alive;
            for (int i = 0; i < maxit; ++i) {
                if (!__any(act)) break;
                {   // gradient 2 Q y + q + 2 rho A_f^T (A_f y + bPk)
                    double z[9], wp[9];
                    UNROLL for (int l = 0; l < 9; ++l) wp[l] = from_prev(ry[l]);  // row-block t-1 (0 for t == 0)
                    UNROLL for (int l = 0; l < 9; ++l) z[l] = ry[l] - wp[l];
                    UNROLL for (int k = 0; k < 3; ++k) z[3 + k] += dtp * wp[k];
                    z[0] += SZ * ry[7] - SY * ry[8];
                    z[1] += SX * ry[8] - SZ * ry[6];
                    z[2] += SY * ry[6] - SX * ry[7];
                    UNROLL for (int l = 0; l < 9; ++l) {
                        const double wi = l0 ? (y[l] + bpi[l]) : 0.0;   // x_init rows
                        g[l] = q2[l] * y[l] + q[l] + rho2 * (z[l] + wi);
                    }
                }
                double Gn;
                bool pend = act;
                do {
                    const double invL = 1.0 / L_x;
                    UNROLL for (int l = 0; l < 9; ++l) {
                        double v = y[l] - g[l] * invL;
                        if (l < NB) v = fmax(fmin(v, ub[l]), lb[l]);   // fista.cpp:10
                        y1[l] = v;
                    }
                    applyA(y1, r1);
                    double g2 = 0, gd = 0, od = 0, n1 = 0, n0 = 0;
                    UNROLL for (int l = 0; l < 9; ++l) {
                        const double d = y1[l] - y[l];
                        g2 += d * d;
                        gd += g[l] * d;
                        od += ((y1[l] + y[l]) * (0.5 * q2[l]) + q[l]) * d;
                        n1 += r1[l] * r1[l];
                        n0 += ry[l] * ry[l];
                        const double i1 = l0 ? (y1[l] + bpi[l]) : 0.0, i0 = l0 ? (y[l] + bpi[l]) : 0.0;
                        n1 += i1 * i1;
                        n0 += i0 * i0;
                    }
                    od += rho * (n1 - n0);
                    g2 = seg_sum<LPP>(g2);
                    gd = seg_sum<LPP>(gd);
                    od = seg_sum<LPP>(od);
                    Gn = sqrt(g2);
                    const bool bt = pend && (od > gd + (L_x * 0.5) * (Gn * Gn));
                    if (bt) { L_x *= beta; ++bt_x; }
                    pend = bt;
                } while (__any(pend));
                const double tk1 = 1.0 + sqrt(1.0 + 4.0 * tk * tk) * 0.5;  // sic
                const double cm = (tk - 1.0) / tk1;
                const bool done = Gn < tol;
                const bool adv = act && !done;
                UNROLL for (int l = 0; l < 9; ++l) {
                    const double yn = y1[l] + cm * (y1[l] - x[l]);
                    const double rn = r1[l] + cm * (r1[l] - rx[l]);
                    x[l] = act ? y1[l] : x[l];
                    rx[l] = act ? r1[l] : rx[l];
                    y[l] = adv ? yn : y[l];
                    ry[l] = adv ? rn : ry[l];
                }
                tk = adv ? tk1 : tk;
                it_x += act ? 1 : 0;
                act = adv;
            }
            if (kvalid) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = x[l]; }

            // dyn_violation = A_f X - b_f ; P += dyn_violation          (biconvex.cpp:98-99)
            double v2 = 0;
            {
                double xn[9], w[9];
                UNROLL for (int l = 0; l < 9; ++l) xn[l] = from_next(x[l]);
                UNROLL for (int l = 0; l < 9; ++l) w[l] = x[l] - xn[l];
                UNROLL for (int k = 0; k < 3; ++k) w[k] += dt * xn[3 + k];
                w[6] += SY * x[2] - SZ * x[1];
                w[7] += SZ * x[0] - SX * x[2];
                w[8] += SX * x[1] - SY * x[0];
                UNROLL for (int l = 0; l < 9; ++l) {
                    const double d = rvalid ? (w[l] - bf[l]) : 0.0;
                    const double di = l0 ? (x[l] - xin[l]) : 0.0;
                    if (alive) { P[l] += d; PI[l] += di; }
                    v2 += d * d + di * di;
                }
            }
            v2 = seg_sum<LPP>(v2);
            const double nrm = sqrt(v2);
            if (alive) {
                last_viol = nrm;
                ++n_admm;
                if (a.hist && l0) a.hist[pb * a.c.num_iters + it] = nrm;
                if (isnan(nrm)) { status = 2; alive = false; }       // biconvex.cpp:106-109
                else if (nrm < exit_tol) alive = false;               // biconvex.cpp:111-114
            }
        }
    }

    // ---- results
    if (rvalid) { UNROLL for (int l = 0; l < 9; ++l) a.P[pb * nx + 9L * t + l] = P[l]; }
    if (l0) {
        UNROLL for (int l = 0; l < 9; ++l) a.P[pb * nx + 9L * H + l] = PI[l];
        a.L_x[pb] = L_x;
        a.L_f[pb] = L_f;
        if (a.dyn_viol) a.dyn_viol[pb] = last_viol;
        if (a.stats) {
            int *s = a.stats + pb * kStats;
            s[0] = n_admm; s[1] = it_f; s[2] = it_x; s[3] = bt_f; s[4] = bt_x; s[5] = status;
        }
    }
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
}

template <int LPP, bool RAW>
hipError_t launch(const BatchArgs &a, hipStream_t stream) {
    const int per_wave = 64 / LPP;
    const unsigned grid = (unsigned)((a.B + per_wave - 1) / per_wave);
    hipLaunchKernelGGL((biconvex_admm_kernel<LPP, 4, RAW>), dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_biconvex_admm(const BatchArgs &a, int n_eff, hipStream_t stream) {
    if (n_eff != 4 || a.H < 1 || a.H + 1 > kMaxKnots || a.B < 0) return hipErrorInvalidValue;
    if (a.B == 0) return hipSuccess;
    const int k = a.H + 1;
    if (k <= 16) return a.raw ? launch<16, true>(a, stream) : launch<16, false>(a, stream);
    if (k <= 32) return a.raw ? launch<32, true>(a, stream) : launch<32, false>(a, stream);
    return a.raw ? launch<64, true>(a, stream) : launch<64, false>(a, stream);
}

hipError_t launch_lane_selftest(const double *in, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(lane_selftest_kernel, dim3(1), dim3(64), 0, stream, in, out);
    return hipGetLastError();
}

const char *biconvex_kernel_name(int H, int raw) {
    (void)raw;
    const int k = H + 1;
    return k <= 16 ? "biconvex_admm_kernel<16" : (k <= 32 ? "biconvex_admm_kernel<32" : "biconvex_admm_kernel<64");
}

}  // namespace bunmpc

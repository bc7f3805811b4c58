// Batched whole-body inverse-kinematics DDP for gfx950 (MI355X).
//
// Restates ik::InverseKinematics::optimize (ISL/src/ik/inverse_kinematics.cpp:54-71): a crocoddyl
// ShootingProblem of IntegratedActionModelEuler nodes around the "kinematic" differential model
// (ISL/src/ik/action_model.cpp:43-94: xout = u, Fx = 0, Fu = I) with the residual costs of
// ISL/src/ik/{com_tasks,end_effector_tasks,regularization_costs}.cpp, solved by
// crocoddyl::SolverDDP::solve() with all defaults.  crocoddyl 1.9.0 / pinocchio 2.6.9 are third
// party and absent: semantics follow oracle/ik_ddp_np.py (PARITY UNPINNED).
//
// MI355X organisation (nothing like crocoddyl's object graph): three kernels per DDP iteration over
// the whole batch, all per-problem state in one contiguous HBM workspace (IkLayout):
//   ik_calcdiff_kernel  one THREAD per (problem, node): kinematics, residuals, Gauss-Newton
//                       L_x / L_xx, Euler Jacobian blocks.  B*(T+1) independent threads.
//   ik_backward_kernel  one WAVE per problem: Riccati recursion with V_xx, Q_xx, Q_xu, Q_uu, K held
//                       in LDS (~50 KB/wave), exploiting F_x = [[A, dt B],[0, I]], F_u = [[dt^2 B],[dt I]]
//                       (A, B identity except a 6x6 free-flyer block) so F^T V F costs O(n^2);
//                       Cholesky and triangular solves cooperative across the 64 lanes;
//                       regularisation retries inside the kernel.
//   ik_forward_kernel   one WAVE per problem: line search 2^-k, k = 0..9 -- lanes 0..17 apply the
//                       feedback u = u - a k - K dx, lane 0 rolls the node model forward; acceptance,
//                       regularisation update and stopping test as crocoddyl 1.9.0 solver-ddp.cpp.
// The host loops over DDP iterations and stops when the device-side active counter reaches zero.
#include "ik_types.h"
#include "rbd_device.h"

namespace bunmpc {
namespace {

using namespace rbd;

struct NodeTasks {
    const double *t;  // kNodeTaskDoubles
    __device__ double frame_w(int s) const { return t[5 * s]; }
    __device__ int frame_id(int s) const { return (int)t[5 * s + 1]; }
    __device__ const double *frame_ref(int s) const { return t + 5 * s + 2; }
    __device__ double com_w() const { return t[5 * kFrameSlots]; }
    __device__ const double *com_ref() const { return t + 5 * kFrameSlots + 1; }
    __device__ double mom_w() const { return t[5 * kFrameSlots + 4]; }
    __device__ const double *mom_ref() const { return t + 5 * kFrameSlots + 5; }
    __device__ double state_w() const { return t[5 * kFrameSlots + 11]; }
    __device__ double ctrl_w() const { return t[5 * kFrameSlots + 12]; }
};

// Cost of one node at (x, u) and, for running nodes, the Euler step.  With DIFF also L_x, L_xx
// (to global memory, row-major 36x36), L_u, diag(L_uu) and the 6x6 blocks of F_x / F_u.
// Running nodes: cost and derivatives scaled by dt (IntegratedActionModelEuler); terminal: unscaled, u = 0.
template <bool DIFF>
__device__ double node_eval(const RobotModelDev &m, const double *x, const double *u, const NodeTasks &tk,
                            const double *state_w, const double *x_reg, const double *ctrl_w, double dt, bool terminal,
                            double *xnext, double *Lx, double *Lxx, double *Lu, double *Luu, double *A6, double *B6) {
    Kin k;
    kin_compute<true>(m, x, k);
    double cost = 0.0;
    // Jacobian storage (DIFF only)
    double Rm[6][kNDX];          // centroidal momentum: [dh/dq, A_g]
    double Jc[3][kNV];           // CoM
    double Jf[kFrameSlots][3][kNV];
    double Jl[36];               // Jlog6 of the state residual's base block
    double rm[6], rc[3], rf[kFrameSlots][3], rs[kNDX];
    const double wm = tk.mom_w(), wc = tk.com_w(), ws = tk.state_w(), wu = tk.ctrl_w();
    const bool has_mom = tk.mom_w() != 0.0 || DIFF;   // weight-0 costs contribute 0 either way

    // ---- residuals
    for (int c = 0; c < 6; ++c) rm[c] = k.hg[c] - tk.mom_ref()[c];
    for (int c = 0; c < 3; ++c) rc[c] = k.com[c] - tk.com_ref()[c];
    cost += wm * 0.5 * (rm[0] * rm[0] + rm[1] * rm[1] + rm[2] * rm[2] + rm[3] * rm[3] + rm[4] * rm[4] + rm[5] * rm[5]);
    cost += wc * 0.5 * dot3(rc, rc);
    for (int s = 0; s < kFrameSlots; ++s) {
        const double w = tk.frame_w(s);
        rf[s][0] = rf[s][1] = rf[s][2] = 0.0;
        if (w != 0.0) {
            double xf[3];
            frame_position(m, k, tk.frame_id(s), xf);
            for (int c = 0; c < 3; ++c) rf[s][c] = xf[c] - tk.frame_ref(s)[c];
            cost += w * 0.5 * dot3(rf[s], rf[s]);
        }
    }
    if (ws != 0.0) {
        state_diff<DIFF>(x_reg, x, rs, Jl);
        double a = 0.0;
        for (int i = 0; i < kNDX; ++i) a += state_w[i] * rs[i] * rs[i];
        cost += ws * 0.5 * a;
    } else {
        for (int i = 0; i < kNDX; ++i) rs[i] = 0.0;
        if (DIFF) { for (int i = 0; i < 36; ++i) Jl[i] = (i % 7 == 0) ? 1.0 : 0.0; }
    }
    if (!terminal && wu != 0.0) {
        double a = 0.0;
        for (int i = 0; i < kNV; ++i) a += ctrl_w[i] * u[i] * u[i];
        cost += wu * 0.5 * a;
    }
    (void)has_mom;

    if (DIFF) {
        // ---- subtree composites / momenta per joint (serial chains), whole robot for the base columns
        Comp call; comp_zero(call);
        for (int b = 0; b <= m.nj; ++b) comp_add(call, k.body[b]);
        for (int col = 0; col < kNV; ++col) {
            Comp cs; double hs[6];
            if (col < 6) { cs = call; for (int c = 0; c < 6; ++c) hs[c] = k.hO[c]; }
            else {
                comp_zero(cs);
                for (int c = 0; c < 6; ++c) hs[c] = 0.0;
                for (int j = col - 6; j <= m.chain_end[col - 6]; ++j) {
                    comp_add(cs, k.body[j + 1]);
                    for (int c = 0; c < 6; ++c) hs[c] += k.hb[j + 1][c];
                }
            }
            const double *S = k.S[col];
            double h[6], t3[3];
            comp_apply(cs, S, h);                       // (l, n_O) of the subtree moved by this column
            for (int c = 0; c < 3; ++c) Jc[c][col] = h[c] / k.M;
            cross3(k.com, h, t3);
            for (int c = 0; c < 3; ++c) { Rm[c][kNV + col] = h[c]; Rm[3 + c][kNV + col] = h[3 + c] - t3[c]; }
            // d h_O / d q_col = S x* h_sub - I_sub (S x V_parent)
            double cf[6], sxv[6], ih[6], a3[3], b3[3];
            cross3(S + 3, hs, cf);
            cross3(S + 3, hs + 3, a3); cross3(S, hs, b3);
            for (int c = 0; c < 3; ++c) cf[3 + c] = a3[c] + b3[c];
            if (col < 6) { for (int c = 0; c < 6; ++c) sxv[c] = 0.0; }
            else {
                const double *Vp = k.V[m.parent[col - 6] + 1];
                cross3(S + 3, Vp, a3); cross3(S, Vp + 3, b3);
                for (int c = 0; c < 3; ++c) sxv[c] = a3[c] + b3[c];
                cross3(S + 3, Vp + 3, sxv + 3);
            }
            comp_apply(cs, sxv, ih);
            double dO[6];
            for (int c = 0; c < 6; ++c) dO[c] = cf[c] - ih[c];
            const double jc[3] = {Jc[0][col], Jc[1][col], Jc[2][col]};
            cross3(jc, k.hO, a3); cross3(k.com, dO, b3);
            for (int c = 0; c < 3; ++c) { Rm[c][col] = dO[c]; Rm[3 + c][col] = dO[3 + c] - a3[c] - b3[c]; }
        }
        for (int s = 0; s < kFrameSlots; ++s) {
            const bool on = tk.frame_w(s) != 0.0;
            double xf[3] = {0, 0, 0};
            int fb = 0;
            if (on) { frame_position(m, k, tk.frame_id(s), xf); fb = m.frame_body[tk.frame_id(s)]; }
            for (int col = 0; col < kNV; ++col) {
                double j3[3] = {0, 0, 0};
                if (on && in_support(m, fb, col)) {
                    cross3(k.S[col] + 3, xf, j3);
                    for (int c = 0; c < 3; ++c) j3[c] += k.S[col][c];
                }
                for (int c = 0; c < 3; ++c) Jf[s][c][col] = j3[c];
            }
        }
        // ---- L_x
        const double sc = terminal ? 1.0 : dt;
        for (int i = 0; i < kNDX; ++i) {
            double g = 0.0;
            for (int r = 0; r < 6; ++r) g += Rm[r][i] * rm[r];
            g *= wm;
            if (i < kNV) {
                g += wc * (Jc[0][i] * rc[0] + Jc[1][i] * rc[1] + Jc[2][i] * rc[2]);
                for (int s = 0; s < kFrameSlots; ++s)
                    g += tk.frame_w(s) * (Jf[s][0][i] * rf[s][0] + Jf[s][1][i] * rf[s][1] + Jf[s][2][i] * rf[s][2]);
            }
            if (i < 6) { double a = 0.0; for (int r = 0; r < 6; ++r) a += Jl[6 * r + i] * state_w[r] * rs[r]; g += ws * a; }
            else g += ws * state_w[i] * rs[i];
            Lx[i] = sc * g;
        }
        // ---- L_xx (Gauss-Newton), symmetric
        for (int i = 0; i < kNDX; ++i)
            for (int j = i; j < kNDX; ++j) {
                double h = 0.0;
                for (int r = 0; r < 6; ++r) h += Rm[r][i] * Rm[r][j];
                h *= wm;
                if (j < kNV) {
                    h += wc * (Jc[0][i] * Jc[0][j] + Jc[1][i] * Jc[1][j] + Jc[2][i] * Jc[2][j]);
                    for (int s = 0; s < kFrameSlots; ++s)
                        h += tk.frame_w(s) * (Jf[s][0][i] * Jf[s][0][j] + Jf[s][1][i] * Jf[s][1][j] + Jf[s][2][i] * Jf[s][2][j]);
                }
                if (j < 6) { double a = 0.0; for (int r = 0; r < 6; ++r) a += Jl[6 * r + i] * state_w[r] * Jl[6 * r + j]; h += ws * a; }
                else if (i == j) h += ws * state_w[i];
                h *= sc;
                Lxx[i * kNDX + j] = h;
                Lxx[j * kNDX + i] = h;
            }
        if (!terminal) {
            for (int i = 0; i < kNV; ++i) { Lu[i] = sc * wu * ctrl_w[i] * u[i]; Luu[i] = sc * wu * ctrl_w[i]; }
        }
    }
    if (!terminal) {
        double dx[kNDX];
        const double *v = x + kNQ;
        for (int i = 0; i < kNV; ++i) { dx[i] = v[i] * dt + u[i] * dt * dt; dx[kNV + i] = u[i] * dt; }
        state_integrate(x, dx, xnext);
        if (DIFF) {
            double dR[9], dp[3];
            exp6(dx, dR, dp);
            act_inv(dR, dp, A6);   // Jintegrate w.r.t. x   (free-flyer block)
            jexp6(dx, B6);         // Jintegrate w.r.t. dx  (free-flyer block)
        }
    }
    return terminal ? cost : dt * cost;
}

__device__ const double *batch_ptr(const double *p, long stride, long b) { return p + stride * b; }

// ------------------------------------------------------------------------------- init ---
__global__ void ik_init_kernel(const IkBatchArgs a) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    // SolverAbstract::setCandidate with empty warm start: xs = state zero (neutral q, v = 0), us = 0
    for (int t = 0; t <= a.T; ++t) {
        double *x = ws + L.xs + (long)t * kNX;
        for (int i = 0; i < kNX; ++i) x[i] = 0.0;
        x[6] = 1.0;
    }
    for (long i = 0; i < (long)a.T * kNV; ++i) ws[L.us + i] = 0.0;
    for (int i = 0; i < kNX; ++i) ws[L.xs_try + i] = a.x0[b * kNX + i];   // xs_try_[0] = x0
    double *s = ws + L.scal;
    s[S_COST] = 0; s[S_XREG] = 1e-9; s[S_D1] = 0; s[S_D2] = 0; s[S_STOP] = 0; s[S_FEAS] = 0; s[S_WASFEAS] = 0;
    s[S_DONE] = 0; s[S_ITERS] = 0; s[S_RECALC] = 1; s[S_STATUS] = 0;
    if (b == 0) *a.active = a.B;
}

// --------------------------------------------------------------------------- calcDiff ---
__global__ __launch_bounds__(64) void ik_calcdiff_kernel(const IkBatchArgs a) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nn = a.T + 1;
    if (id >= (long)a.B * nn) return;
    const long b = id / nn;
    const int t = (int)(id % nn);
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    if (ws[L.scal + S_DONE] != 0.0 || ws[L.scal + S_RECALC] == 0.0) return;
    const bool terminal = t == a.T;
    NodeTasks tk{a.tasks + (b * nn + t) * kNodeTaskDoubles};
    double xnext[kNX], Lx[kNDX], Lu[kNV], Luu[kNV], A6[36], B6[36];
    const double dt = terminal ? 0.0 : a.dt[b * a.T + t];
    const double c = node_eval<true>(*a.model, ws + L.xs + (long)t * kNX, terminal ? nullptr : ws + L.us + (long)t * kNV, tk,
                                     batch_ptr(a.state_w, a.s_state_w, b), a.x_reg + b * kNX, batch_ptr(a.ctrl_w, a.s_ctrl_w, b),
                                     dt, terminal, xnext, Lx, ws + L.Lxx + (long)t * kNDX * kNDX, Lu, Luu, A6, B6);
    for (int i = 0; i < kNDX; ++i) ws[L.Lx + (long)t * kNDX + i] = Lx[i];
    // node costs are summed by the backward kernel: parked in the (not yet used) fs slot of this node
    ws[L.fs + (long)t * kNDX] = c;
    if (!terminal) {
        for (int i = 0; i < kNV; ++i) { ws[L.Lu + (long)t * kNV + i] = Lu[i]; ws[L.Luu + (long)t * kNV + i] = Luu[i]; }
        for (int i = 0; i < 36; ++i) { ws[L.A6 + (long)t * 36 + i] = A6[i]; ws[L.B6 + (long)t * 36 + i] = B6[i]; }
        for (int i = 0; i < kNX; ++i) ws[L.xnext + (long)t * kNX + i] = xnext[i];
    }
}

// --------------------------------------------------------------------------- backward ---
constexpr int LD = kNDX + 1;   // padded leading dimension of the 36-wide LDS matrices
constexpr int LDU = kNV + 1;

struct BackwardLds {
    double V[kNDX * LD], M1[kNDX * LD], W[kNDX * LD];
    double Qxu[kNDX * LDU], VFu[kNDX * LDU], Kt[kNV * LD], Quu[kNV * LDU];
    double Vx[kNDX], Qx[kNDX], Qu[kNV], kf[kNV], fs[kNDX], A6[36], B6[36], Luu[kNV], tmp[kNDX];
    int flag;
};

__global__ __launch_bounds__(64) void ik_backward_kernel(const IkBatchArgs a) {
    __shared__ BackwardLds s;
    const long b = blockIdx.x;
    const int lane = threadIdx.x;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    double *sc = ws + L.scal;
    if (sc[S_DONE] != 0.0) return;
    const int T = a.T;
    bool feas = sc[S_FEAS] != 0.0;
    const bool wasfeas = sc[S_WASFEAS] != 0.0;

    if (sc[S_RECALC] != 0.0) {
        // SolverDDP::calcDiff tail: total cost and the gaps fs (solver-ddp.cpp calcDiff)
        __syncthreads();
        double c = 0.0;
        if (lane == 0) { for (int t = 0; t <= T; ++t) c += ws[L.fs + (long)t * kNDX]; sc[S_COST] = c; }
        __syncthreads();
        if (!feas) {
            double mx = 0.0;
            if (lane <= T) {
                double d[kNDX];
                const double *xa = ws + L.xs + (long)lane * kNX;
                const double *xb = lane == 0 ? a.x0 + b * kNX : ws + L.xnext + (long)(lane - 1) * kNX;
                state_diff<false>(xa, xb, d, nullptr);
                for (int i = 0; i < kNDX; ++i) { ws[L.fs + (long)lane * kNDX + i] = d[i]; mx = fmax(mx, fabs(d[i])); }
            }
            const bool ok = __all(mx < 1e-16);   // th_gaptol_
            feas = ok;
            if (lane == 0) sc[S_FEAS] = ok ? 1.0 : 0.0;
        } else if (!wasfeas) {
            for (long i = lane; i < (long)(T + 1) * kNDX; i += 64) ws[L.fs + i] = 0.0;
        } else {
            // fs slots were used to park node costs: restore zeros
            if (lane <= T) ws[L.fs + (long)lane * kNDX] = 0.0;
        }
        __syncthreads();
    }

    double xreg = sc[S_XREG];
    for (;;) {   // computeDirection with regularisation retries (solver-ddp.cpp solve())
        if (lane == 0) s.flag = 0;
        // terminal node
        for (int e = lane; e < kNDX * kNDX; e += 64) {
            const int i = e / kNDX, j = e % kNDX;
            s.V[i * LD + j] = ws[L.Lxx + (long)T * kNDX * kNDX + e] + (i == j ? xreg : 0.0);
        }
        if (lane < kNDX) { s.Vx[lane] = ws[L.Lx + (long)T * kNDX + lane]; s.fs[lane] = ws[L.fs + (long)T * kNDX + lane]; }
        __syncthreads();
        if (!feas && lane < kNDX) {
            double acc = 0.0;
            for (int j = 0; j < kNDX; ++j) acc += s.V[lane * LD + j] * s.fs[j];
            s.tmp[lane] = s.Vx[lane] + acc;
        }
        __syncthreads();
        if (!feas && lane < kNDX) s.Vx[lane] = s.tmp[lane];
        __syncthreads();

        for (int t = T - 1; t >= 0; --t) {
            const double dt = a.dt[b * T + t];
            const double dt2 = dt * dt;
            if (lane < 36) { s.A6[lane] = ws[L.A6 + (long)t * 36 + lane]; s.B6[lane] = ws[L.B6 + (long)t * 36 + lane]; }
            if (lane < kNV) s.Luu[lane] = ws[L.Luu + (long)t * kNV + lane];
            if (lane < kNDX) s.fs[lane] = ws[L.fs + (long)t * kNDX + lane];
            for (int e = lane; e < kNDX * kNDX; e += 64) s.W[(e / kNDX) * LD + e % kNDX] = ws[L.Lxx + (long)t * kNDX * kNDX + e];
            __syncthreads();
            // M1 = Fx^T V
            for (int e = lane; e < kNDX * kNDX; e += 64) {
                const int i = e / kNDX, j = e % kNDX;
                double v;
                if (i < 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.A6[6 * c + i] * s.V[c * LD + j]; }
                else if (i < kNV) v = s.V[i * LD + j];
                else if (i < kNV + 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.B6[6 * c + (i - kNV)] * s.V[c * LD + j]; v = dt * v + s.V[i * LD + j]; }
                else v = dt * s.V[(i - kNV) * LD + j] + s.V[i * LD + j];
                s.M1[i * LD + j] = v;
            }
            // Qx = Lx + Fx^T Vx ; Qu = Lu + Fu^T Vx
            if (lane < kNDX) {
                const int i = lane;
                double v;
                if (i < 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.A6[6 * c + i] * s.Vx[c]; }
                else if (i < kNV) v = s.Vx[i];
                else if (i < kNV + 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.B6[6 * c + (i - kNV)] * s.Vx[c]; v = dt * v + s.Vx[i]; }
                else v = dt * s.Vx[i - kNV] + s.Vx[i];
                s.Qx[i] = ws[L.Lx + (long)t * kNDX + i] + v;
            }
            if (lane < kNV) {
                const int q = lane;
                double v;
                if (q < 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.B6[6 * c + q] * s.Vx[c]; } else v = s.Vx[q];
                s.Qu[q] = ws[L.Lu + (long)t * kNV + q] + dt2 * v + dt * s.Vx[kNV + q];
            }
            __syncthreads();
            // W = Lxx + M1 Fx ; Qxu = M1 Fu ; VFu = V Fu
            for (int e = lane; e < kNDX * kNDX; e += 64) {
                const int i = e / kNDX, j = e % kNDX;
                double v;
                if (j < 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.M1[i * LD + c] * s.A6[6 * c + j]; }
                else if (j < kNV) v = s.M1[i * LD + j];
                else if (j < kNV + 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.M1[i * LD + c] * s.B6[6 * c + (j - kNV)]; v = dt * v + s.M1[i * LD + j]; }
                else v = dt * s.M1[i * LD + (j - kNV)] + s.M1[i * LD + j];
                s.W[i * LD + j] += v;
            }
            for (int e = lane; e < kNDX * kNV; e += 64) {
                const int i = e / kNV, q = e % kNV;
                double v1, v2;
                if (q < 6) {
                    v1 = 0.0; v2 = 0.0;
                    for (int c = 0; c < 6; ++c) { v1 += s.M1[i * LD + c] * s.B6[6 * c + q]; v2 += s.V[i * LD + c] * s.B6[6 * c + q]; }
                } else { v1 = s.M1[i * LD + q]; v2 = s.V[i * LD + q]; }
                s.Qxu[i * LDU + q] = dt2 * v1 + dt * s.M1[i * LD + kNV + q];
                s.VFu[i * LDU + q] = dt2 * v2 + dt * s.V[i * LD + kNV + q];
            }
            __syncthreads();
            // Quu = Luu + Fu^T (V Fu) + ureg I
            for (int e = lane; e < kNV * kNV; e += 64) {
                const int p = e / kNV, q = e % kNV;
                double v;
                if (p < 6) { v = 0.0; for (int c = 0; c < 6; ++c) v += s.B6[6 * c + p] * s.VFu[c * LDU + q]; } else v = s.VFu[p * LDU + q];
                v = dt2 * v + dt * s.VFu[(kNV + p) * LDU + q];
                if (p == q) v += s.Luu[p] + xreg;
                s.Quu[p * LDU + q] = v;
            }
            __syncthreads();
            // Cholesky (lower, in place); a non-positive or NaN pivot fails the pass (Eigen::LLT info != Success)
            for (int j = 0; j < kNV; ++j) {
                const double piv = s.Quu[j * LDU + j];
                if (!(piv > 0.0)) { if (lane == 0) s.flag = 1; }
                const double d = sqrt(piv);
                __syncthreads();
                if (lane == 0) s.Quu[j * LDU + j] = d;
                if (lane > j && lane < kNV) s.Quu[lane * LDU + j] /= d;
                __syncthreads();
                for (int e = lane; e < kNV * kNV; e += 64) {
                    const int p = e / kNV, q = e % kNV;
                    if (q > j && p >= q) s.Quu[p * LDU + q] -= s.Quu[p * LDU + j] * s.Quu[q * LDU + j];
                }
                __syncthreads();
            }
            // K = Quu^-1 Qxu^T (one right-hand side per lane), k = Quu^-1 Qu (lane 36)
            if (lane <= kNDX) {
                double y[kNV];
                for (int p = 0; p < kNV; ++p) {
                    double v = lane < kNDX ? s.Qxu[lane * LDU + p] : s.Qu[p];
                    for (int q = 0; q < p; ++q) v -= s.Quu[p * LDU + q] * y[q];
                    y[p] = v / s.Quu[p * LDU + p];
                }
                for (int p = kNV - 1; p >= 0; --p) {
                    double v = y[p];
                    for (int q = p + 1; q < kNV; ++q) v -= s.Quu[q * LDU + p] * y[q];
                    y[p] = v / s.Quu[p * LDU + p];
                }
                if (lane < kNDX) { for (int p = 0; p < kNV; ++p) s.Kt[p * LD + lane] = y[p]; }
                else { for (int p = 0; p < kNV; ++p) s.kf[p] = y[p]; }
            }
            __syncthreads();
            // Quuk = Quu k = L (L^T k)
            if (lane < kNV) { double v = 0.0; for (int q = lane; q < kNV; ++q) v += s.Quu[q * LDU + lane] * s.kf[q]; s.tmp[lane] = v; }
            __syncthreads();
            if (lane < kNV) {
                double v = 0.0;
                for (int q = 0; q <= lane; ++q) v += s.Quu[lane * LDU + q] * s.tmp[q];
                ws[L.Quuk + (long)t * kNV + lane] = v;
                ws[L.kff + (long)t * kNV + lane] = s.kf[lane];
                ws[L.Qu + (long)t * kNV + lane] = s.Qu[lane];
            }
            for (int e = lane; e < kNV * kNDX; e += 64) ws[L.K + (long)t * kNV * kNDX + e] = s.Kt[(e / kNDX) * LD + e % kNDX];
            // Vx = Qx - K^T Qu ; Vxx = Qxx - Qxu K (into M1), then symmetrise + xreg
            if (lane < kNDX) { double v = s.Qx[lane]; for (int p = 0; p < kNV; ++p) v -= s.Kt[p * LD + lane] * s.Qu[p]; s.Vx[lane] = v; }
            for (int e = lane; e < kNDX * kNDX; e += 64) {
                const int i = e / kNDX, j = e % kNDX;
                double v = s.W[i * LD + j];
                for (int p = 0; p < kNV; ++p) v -= s.Qxu[i * LDU + p] * s.Kt[p * LD + j];
                s.M1[i * LD + j] = v;
            }
            __syncthreads();
            for (int e = lane; e < kNDX * kNDX; e += 64) {
                const int i = e / kNDX, j = e % kNDX;
                s.V[i * LD + j] = 0.5 * (s.M1[i * LD + j] + s.M1[j * LD + i]) + (i == j ? xreg : 0.0);
            }
            __syncthreads();
            if (!feas && lane < kNDX) { double acc = 0.0; for (int j = 0; j < kNDX; ++j) acc += s.V[lane * LD + j] * s.fs[j]; s.tmp[lane] = s.Vx[lane] + acc; }
            __syncthreads();
            if (!feas && lane < kNDX) s.Vx[lane] = s.tmp[lane];
            // raiseIfNaN on Vx / Vxx
            bool bad = false;
            if (lane < kNDX) bad = !(fabs(s.Vx[lane]) < INFINITY);
            for (int e = lane; e < kNDX * kNDX; e += 64) bad = bad || !(fabs(s.V[(e / kNDX) * LD + e % kNDX]) < INFINITY);
            if (__any(bad) && lane == 0) s.flag = 1;
            __syncthreads();
            if (s.flag) break;
        }
        __syncthreads();
        if (!s.flag) break;
        // increaseRegularization; give up at reg_max (solve() returns false)
        xreg = fmin(xreg * 10.0, 1e9);
        if (lane == 0) { sc[S_XREG] = xreg; sc[S_RECALC] = 0.0; }
        if (xreg == 1e9) {
            if (lane == 0) { sc[S_DONE] = 1.0; sc[S_STATUS] = 2.0; atomicSub(a.active, 1); }
            return;
        }
        __syncthreads();
    }
    // expectedImprovement / stoppingCriteria ingredients
    if (lane == 0) {
        double d1 = 0.0, d2 = 0.0, st = 0.0;
        for (int t = 0; t < T; ++t)
            for (int p = 0; p < kNV; ++p) {
                const double qu = ws[L.Qu + (long)t * kNV + p], kk = ws[L.kff + (long)t * kNV + p];
                d1 += qu * kk; d2 -= kk * ws[L.Quuk + (long)t * kNV + p]; st += qu * qu;
            }
        sc[S_D1] = d1; sc[S_D2] = d2; sc[S_STOP] = st;
    }
}

// ---------------------------------------------------------------------------- forward ---
struct ForwardLds { double dx[kNDX], u[kNV], x[kNX]; double bc[4]; };

__global__ __launch_bounds__(64) void ik_forward_kernel(const IkBatchArgs a) {
    __shared__ ForwardLds s;
    const long b = blockIdx.x;
    const int lane = threadIdx.x;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    double *sc = ws + L.scal;
    if (sc[S_DONE] != 0.0) return;
    const int T = a.T, nn = a.T + 1;
    const double cost = sc[S_COST], d1 = sc[S_D1], d2 = sc[S_D2];
    const bool feas = sc[S_FEAS] != 0.0;
    const double *state_w = batch_ptr(a.state_w, a.s_state_w, b), *ctrl_w = batch_ptr(a.ctrl_w, a.s_ctrl_w, b);
    const double *x_reg = a.x_reg + b * kNX;
    bool accepted = false;
    double alpha = 1.0, cost_try = 0.0;
    for (int ia = 0; ia < 10; ++ia) {   // alphas_ = 2^-n, n = 0..9
        alpha = ldexp(1.0, -ia);
        bool failed = false;
        cost_try = 0.0;
        for (int t = 0; t < T && !failed; ++t) {
            if (lane == 0) {
                state_diff<false>(ws + L.xs + (long)t * kNX, ws + L.xs_try + (long)t * kNX, s.dx, nullptr);
                for (int i = 0; i < kNX; ++i) s.x[i] = ws[L.xs_try + (long)t * kNX + i];
            }
            __syncthreads();
            if (lane < kNV) {
                const double *Kr = ws + L.K + (long)t * kNV * kNDX + (long)lane * kNDX;
                double v = ws[L.us + (long)t * kNV + lane] - alpha * ws[L.kff + (long)t * kNV + lane];
                for (int j = 0; j < kNDX; ++j) v -= Kr[j] * s.dx[j];
                s.u[lane] = v;
                ws[L.us_try + (long)t * kNV + lane] = v;
            }
            __syncthreads();
            if (lane == 0) {
                NodeTasks tk{a.tasks + (b * nn + t) * kNodeTaskDoubles};
                double xn[kNX];
                const double c = node_eval<false>(*a.model, s.x, s.u, tk, state_w, x_reg, ctrl_w, a.dt[b * T + t], false, xn,
                                                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
                bool bad = !(fabs(c) < INFINITY);
                for (int i = 0; i < kNX; ++i) { ws[L.xs_try + (long)(t + 1) * kNX + i] = xn[i]; bad = bad || !(fabs(xn[i]) < INFINITY); }
                s.bc[0] = c; s.bc[1] = bad ? 1.0 : 0.0;
            }
            __syncthreads();
            cost_try += s.bc[0];
            failed = s.bc[1] != 0.0;
            __syncthreads();
        }
        if (!failed) {
            if (lane == 0) {
                NodeTasks tk{a.tasks + (b * nn + T) * kNodeTaskDoubles};
                const double c = node_eval<false>(*a.model, ws + L.xs_try + (long)T * kNX, nullptr, tk, state_w, x_reg, ctrl_w, 0.0,
                                                  true, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
                s.bc[0] = c; s.bc[1] = !(fabs(c) < INFINITY) ? 1.0 : 0.0;
            }
            __syncthreads();
            cost_try += s.bc[0];
            failed = s.bc[1] != 0.0;
            __syncthreads();
        }
        if (failed) continue;   // tryStep threw: next step length
        const double dV = cost - cost_try;
        const double dVexp = alpha * (d1 + 0.5 * alpha * d2);
        if (dVexp >= 0.0 && (d1 < 1e-12 || !feas || dV > 0.1 * dVexp)) { accepted = true; break; }
    }
    double xreg = sc[S_XREG];
    if (accepted) {   // setCandidate(xs_try, us_try, true)
        for (long i = lane; i < (long)nn * kNX; i += 64) ws[L.xs + i] = ws[L.xs_try + i];
        for (long i = lane; i < (long)T * kNV; i += 64) ws[L.us + i] = ws[L.us_try + i];
    }
    if (alpha > 0.5) xreg = fmax(xreg / 10.0, 1e-9);          // decreaseRegularization
    bool done = false;
    double status = 0.0;
    if (alpha <= 0.01) {                                         // increaseRegularization
        xreg = fmin(xreg * 10.0, 1e9);
        if (xreg == 1e9) { done = true; status = 2.0; }
    }
    const bool wasfeas_new = accepted ? feas : (sc[S_WASFEAS] != 0.0);
    const double iters = sc[S_ITERS] + 1.0;
    if (!done && wasfeas_new && sc[S_STOP] < 1e-9) { done = true; status = 0.0; }       // converged
    if (!done && iters >= (double)a.maxiter) { done = true; status = 1.0; }             // maxiter reached
    __syncthreads();
    if (lane == 0) {
        if (accepted) { sc[S_WASFEAS] = feas ? 1.0 : 0.0; sc[S_FEAS] = 1.0; sc[S_COST] = cost_try; sc[S_RECALC] = 1.0; }
        else sc[S_RECALC] = 0.0;
        sc[S_XREG] = xreg; sc[S_ITERS] = iters;
        if (done) { sc[S_DONE] = 1.0; sc[S_STATUS] = status; atomicSub(a.active, 1); }
    }
}

// ------------------------------------------------------------ small helper kernels ---
__global__ void ik_centroidal_state_kernel(const RobotModelDev *model, const double *x, double *out9, int B) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    Kin k;
    kin_compute<true>(*model, x + b * kNX, k);
    for (int c = 0; c < 3; ++c) {
        out9[b * 9 + c] = k.com[c];
        out9[b * 9 + 3 + c] = k.hg[c] / k.M;     // vcom
        out9[b * 9 + 6 + c] = k.hg[3 + c];       // hg.angular
    }
}

__global__ void ik_com_mom_kernel(const RobotModelDev *model, const double *xs, double *com, double *mom, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Kin k;
    kin_compute<true>(*model, xs + i * kNX, k);
    for (int c = 0; c < 3; ++c) com[i * 3 + c] = k.com[c];
    for (int c = 0; c < 6; ++c) mom[i * 6 + c] = k.hg[c];
}

// com / momentum references of the IK tracking tasks from the centroidal solution X
// (KinoDynMP::optimize, kino_dyn.cpp:50-56: rows 0..T-1 running, row T terminal; mom = [m v, L])
__global__ void kd_fill_refs_kernel(double *tasks, const double *X, double m, int B, int H, int T) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * (T + 1)) return;
    const long b = id / (T + 1);
    const int t = (int)(id % (T + 1));
    double *tk = tasks + id * kNodeTaskDoubles;
    const double *Xk = X + b * 9L * (H + 1) + 9L * t;
    for (int c = 0; c < 3; ++c) {
        tk[5 * kFrameSlots + 1 + c] = Xk[c];
        tk[5 * kFrameSlots + 5 + c] = m * Xk[3 + c];
        tk[5 * kFrameSlots + 8 + c] = Xk[6 + c];
    }
}

}  // namespace

hipError_t ik_launch_fill_refs(double *tasks, const double *X, double m, int B, int H, int T, hipStream_t st) {
    const long n = (long)B * (T + 1);
    hipLaunchKernelGGL(kd_fill_refs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, tasks, X, m, B, H, T);
    return hipGetLastError();
}

hipError_t ik_launch_init(const IkBatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ik_init_kernel, dim3((a.B + 63) / 64), dim3(64), 0, st, a);
    return hipGetLastError();
}
hipError_t ik_launch_calcdiff(const IkBatchArgs &a, hipStream_t st) {
    const long n = (long)a.B * (a.T + 1);
    hipLaunchKernelGGL(ik_calcdiff_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
    return hipGetLastError();
}
hipError_t ik_launch_backward(const IkBatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ik_backward_kernel, dim3(a.B), dim3(64), 0, st, a);
    return hipGetLastError();
}
hipError_t ik_launch_forward(const IkBatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ik_forward_kernel, dim3(a.B), dim3(64), 0, st, a);
    return hipGetLastError();
}
hipError_t ik_launch_centroidal_state(const RobotModelDev *model, const double *x, double *out9, int B, hipStream_t st) {
    hipLaunchKernelGGL(ik_centroidal_state_kernel, dim3((B + 63) / 64), dim3(64), 0, st, model, x, out9, B);
    return hipGetLastError();
}
hipError_t ik_launch_com_mom(const RobotModelDev *model, const double *xs, double *com, double *mom, int n, hipStream_t st) {
    hipLaunchKernelGGL(ik_com_mom_kernel, dim3((n + 63) / 64), dim3(64), 0, st, model, xs, com, mom, n);
    return hipGetLastError();
}

}  // namespace bunmpc

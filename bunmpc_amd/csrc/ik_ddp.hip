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
//   ik_calcdiff_kernel  one WAVE per (problem, node): every lane runs the register-resident robot
//                       pass (rbd_quad.h), lanes 0..17 each own one velocity column (CoM Jacobian,
//                       A_g, dh_g/dq, frame Jacobians) written to LDS, then all 64 lanes assemble
//                       the Gauss-Newton L_x / L_xx from those rows (coalesced 10 KB store).
//   ik_backward_kernel  one WAVE per problem: Riccati recursion with V_xx, Q_xx, Q_xu, Q_uu, K held
//                       in LDS (~53 KB/wave), exploiting F_x = [[A, dt B],[0, I]], F_u = [[dt^2 B],[dt I]]
//                       (A, B identity except a 6x6 free-flyer block) so F^T V F costs O(n^2);
//                       Cholesky and triangular solves cooperative across the 64 lanes, in LDS;
//                       regularisation retries inside the kernel.
//   ik_forward_kernel   one WAVE per problem: line search 2^-k, k = 0..9 -- lanes 0..17 apply the
//                       feedback u = u - a k - K dx, then the node cost / Euler step; acceptance,
//                       regularisation update and stopping test as crocoddyl 1.9.0 solver-ddp.cpp.
// The host loops over DDP iterations and stops when the device-side active counter reaches zero.
#include "ik_types.h"
#include "rbd_quad.h"

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

struct Residuals { double rm[6], rc[3], rf[kFrameSlots][3]; };

// pass 1 + the residuals of the momentum / CoM / frame costs and their cost (unscaled, without the
// state and control terms)
template <bool COMPOSITE>
__device__ __forceinline__ double kin_costs(const RobotModelDev &m, const double *x, const NodeTasks &tk, Pass1 &p1, Residuals &r) {
    int fid[kFrameSlots];
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) fid[s] = tk.frame_w(s) != 0.0 ? tk.frame_id(s) : -1;
    quad_pass1<COMPOSITE>(m, x, fid, p1);
    double cost = 0.0, a = 0.0;
    UNROLL_RBD for (int c = 0; c < 6; ++c) { r.rm[c] = p1.hg[c] - tk.mom_ref()[c]; a += r.rm[c] * r.rm[c]; }
    cost += tk.mom_w() * 0.5 * a;
    UNROLL_RBD for (int c = 0; c < 3; ++c) r.rc[c] = p1.com[c] - tk.com_ref()[c];
    cost += tk.com_w() * 0.5 * dot3(r.rc, r.rc);
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) {
        const double w = tk.frame_w(s);
        UNROLL_RBD for (int c = 0; c < 3; ++c) r.rf[s][c] = w != 0.0 ? p1.fx[s][c] - tk.frame_ref(s)[c] : 0.0;
        cost += w * 0.5 * dot3(r.rf[s], r.rf[s]);
    }
    return cost;
}

// IntegratedActionModelEuler: dx = [v dt + u dt^2 ; u dt], xnext = x (+) dx; optionally the 6x6 blocks of
// Jintegrate (w.r.t. x: A6, w.r.t. dx: B6)
template <bool JAC>
__device__ __forceinline__ void euler_step(const double *x, const double *u, double dt, double *xnext, double *A6, double *B6) {
    double dx[kNDX];
    const double *v = x + kNQ;
    UNROLL_RBD for (int i = 0; i < kNV; ++i) { dx[i] = v[i] * dt + u[i] * dt * dt; dx[kNV + i] = u[i] * dt; }
    state_integrate(x, dx, xnext);
    if (JAC) {
        double dR[9], dp[3];
        exp6(dx, dR, dp);
        act_inv(dR, dp, A6);
        jexp6(dx, B6);
    }
}

// node cost at (x, u) and the Euler step (forward pass); x, u may live in LDS
__device__ __forceinline__ double node_cost(const RobotModelDev &m, const double *x, const double *u, const NodeTasks &tk, const double *state_w,
                            const double *x_reg, const double *ctrl_w, double dt, bool terminal, double *xnext) {
    Pass1 p1; Residuals r;
    double cost = kin_costs<false>(m, x, tk, p1, r);
    if (tk.state_w() != 0.0) {
        double rs[kNDX], a = 0.0;
        state_diff<false>(x_reg, x, rs, nullptr);
        UNROLL_RBD for (int i = 0; i < kNDX; ++i) a += state_w[i] * rs[i] * rs[i];
        cost += tk.state_w() * 0.5 * a;
    }
    if (terminal) return cost;
    if (tk.ctrl_w() != 0.0) {
        double a = 0.0;
        UNROLL_RBD for (int i = 0; i < kNV; ++i) a += ctrl_w[i] * u[i] * u[i];
        cost += tk.ctrl_w() * 0.5 * a;
    }
    euler_step<false>(x, u, dt, xnext, nullptr, nullptr);
    return dt * cost;
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
struct CalcLds {
    RobotModelDev m;
    double x[kNX], u[kNV];
    double Rm[6][kNDX];               // centroidal momentum rows [dh/dq, A_g]
    double Jc[3][kNV];                // CoM rows
    double Jf[kFrameSlots][3][kNV];   // frame rows
    double rs[kNDX], Jl[36];          // state residual and the Jlog6 block of its Jacobian
};

__global__ __launch_bounds__(64) void ik_calcdiff_kernel(const IkBatchArgs a) {
    __shared__ CalcLds s;
    const int nn = a.T + 1;
    const long b = blockIdx.x / nn;
    const int t = blockIdx.x % nn, lane = threadIdx.x;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    if (ws[L.scal + S_DONE] != 0.0 || ws[L.scal + S_RECALC] == 0.0) return;
    const bool terminal = t == a.T;
    {
        const int *src = reinterpret_cast<const int *>(a.model);
        int *dst = reinterpret_cast<int *>(&s.m);
        for (int i = lane; i < (int)(sizeof(RobotModelDev) / sizeof(int)); i += 64) dst[i] = src[i];
    }
    const RobotModelDev &m = s.m;
    NodeTasks tk{a.tasks + (b * nn + t) * kNodeTaskDoubles};
    const double *state_w = batch_ptr(a.state_w, a.s_state_w, b), *ctrl_w = batch_ptr(a.ctrl_w, a.s_ctrl_w, b);
    const double dt = terminal ? 0.0 : a.dt[b * a.T + t];
    if (lane < kNX) s.x[lane] = ws[L.xs + (long)t * kNX + lane];
    if (lane < kNV) s.u[lane] = terminal ? 0.0 : ws[L.us + (long)t * kNV + lane];
    __syncthreads();
    Pass1 p1; Residuals r;
    double cost = kin_costs<true>(m, s.x, tk, p1, r);     // every lane: the same robot pass, in registers
    const double wm = tk.mom_w(), wc = tk.com_w(), wst = tk.state_w(), wu = tk.ctrl_w();
    if (lane < kNV) {   // one velocity column per lane
        Column c;
        quad_column(m, s.x, p1, lane, c);
        UNROLL_RBD for (int k = 0; k < 3; ++k) s.Jc[k][lane] = c.jc[k];
        UNROLL_RBD for (int k = 0; k < 6; ++k) { s.Rm[k][lane] = c.dh[k]; s.Rm[k][kNV + lane] = c.ag[k]; }
        UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) {
            double j3[3] = {0, 0, 0};
            if (tk.frame_w(f) != 0.0 && quad_supports(m, tk.frame_id(f), lane)) {
                cross3(c.S + 3, p1.fx[f], j3);
                UNROLL_RBD for (int k = 0; k < 3; ++k) j3[k] += c.S[k];
            }
            UNROLL_RBD for (int k = 0; k < 3; ++k) s.Jf[f][k][lane] = j3[k];
        }
    } else if (lane == 32) {   // state residual + its Jacobian block, Euler step, node cost
        double rs[kNDX], Jl[36];
        if (wst != 0.0) {
            state_diff<true>(a.x_reg + b * kNX, s.x, rs, Jl);
            double acc = 0.0;
            UNROLL_RBD for (int i = 0; i < kNDX; ++i) acc += state_w[i] * rs[i] * rs[i];
            cost += wst * 0.5 * acc;
        } else {
            UNROLL_RBD for (int i = 0; i < kNDX; ++i) rs[i] = 0.0;
            UNROLL_RBD for (int i = 0; i < 36; ++i) Jl[i] = (i % 7 == 0) ? 1.0 : 0.0;
        }
        UNROLL_RBD for (int i = 0; i < kNDX; ++i) s.rs[i] = rs[i];
        UNROLL_RBD for (int i = 0; i < 36; ++i) s.Jl[i] = Jl[i];
        if (!terminal) {
            double acc = 0.0;
            UNROLL_RBD for (int i = 0; i < kNV; ++i) acc += ctrl_w[i] * s.u[i] * s.u[i];
            cost += wu * 0.5 * acc;
            double xn[kNX], A6[36], B6[36];
            euler_step<true>(s.x, s.u, dt, xn, A6, B6);
            UNROLL_RBD for (int i = 0; i < kNX; ++i) ws[L.xnext + (long)t * kNX + i] = xn[i];
            UNROLL_RBD for (int i = 0; i < 36; ++i) { ws[L.A6 + (long)t * 36 + i] = A6[i]; ws[L.B6 + (long)t * 36 + i] = B6[i]; }
            cost *= dt;
        }
        // node costs are summed by the backward kernel: parked in the fs slot of this node
        ws[L.fs + (long)t * kNDX] = cost;
    }
    __syncthreads();
    const double sc = terminal ? 1.0 : dt;
    // L_x
    if (lane < kNDX) {
        const int i = lane;
        double g = 0.0;
        UNROLL_RBD for (int k = 0; k < 6; ++k) g += s.Rm[k][i] * r.rm[k];
        g *= wm;
        if (i < kNV) {
            g += wc * (s.Jc[0][i] * r.rc[0] + s.Jc[1][i] * r.rc[1] + s.Jc[2][i] * r.rc[2]);
            UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
                g += tk.frame_w(f) * (s.Jf[f][0][i] * r.rf[f][0] + s.Jf[f][1][i] * r.rf[f][1] + s.Jf[f][2][i] * r.rf[f][2]);
        }
        if (i < 6) { double acc = 0.0; for (int k = 0; k < 6; ++k) acc += s.Jl[6 * k + i] * state_w[k] * s.rs[k]; g += wst * acc; }
        else g += wst * state_w[i] * s.rs[i];
        ws[L.Lx + (long)t * kNDX + i] = sc * g;
    }
    // L_xx (Gauss-Newton): every lane ~20 entries, rows read from LDS, coalesced store
    double *Lxx = ws + L.Lxx + (long)t * kNDX * kNDX;
    double fw[kFrameSlots];
    UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) fw[f] = tk.frame_w(f);
    for (int e = lane; e < kNDX * kNDX; e += 64) {
        const int i = e / kNDX, j = e % kNDX;
        double h = 0.0;
        UNROLL_RBD for (int k = 0; k < 6; ++k) h += s.Rm[k][i] * s.Rm[k][j];
        h *= wm;
        if (i < kNV && j < kNV) {
            h += wc * (s.Jc[0][i] * s.Jc[0][j] + s.Jc[1][i] * s.Jc[1][j] + s.Jc[2][i] * s.Jc[2][j]);
            UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
                h += fw[f] * (s.Jf[f][0][i] * s.Jf[f][0][j] + s.Jf[f][1][i] * s.Jf[f][1][j] + s.Jf[f][2][i] * s.Jf[f][2][j]);
        }
        if (i < 6 && j < 6) { double acc = 0.0; for (int k = 0; k < 6; ++k) acc += s.Jl[6 * k + i] * state_w[k] * s.Jl[6 * k + j]; h += wst * acc; }
        else if (i == j) h += wst * state_w[i];
        Lxx[e] = sc * h;
    }
    if (!terminal && lane < kNV) {
        ws[L.Lu + (long)t * kNV + lane] = sc * wu * ctrl_w[lane] * s.u[lane];
        ws[L.Luu + (long)t * kNV + lane] = sc * wu * ctrl_w[lane];
    }
}

// --------------------------------------------------------------------------- backward ---
constexpr int LD = kNDX + 1;   // padded leading dimension of the 36-wide LDS matrices
constexpr int LDU = kNV + 1;
constexpr int kBwdThreads = 256;   // four waves share one problem's Riccati step

struct BackwardLds {
    double V[kNDX * LD], M1[kNDX * LD], W[kNDX * LD];
    double Qxu[kNDX * LDU], VFu[kNDX * LDU], Kt[kNV * LD], Quu[kNV * LDU];
    double Vx[kNDX], Qx[kNDX], Qu[kNV], kf[kNV], fs[kNDX], A6[36], B6[36], Luu[kNV], tmp[kNDX];
    double idg[kNV];
    int flag;
};

__global__ __launch_bounds__(kBwdThreads) void ik_backward_kernel(const IkBatchArgs a) {
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
        // SolverDDP::calcDiff tail: total cost and the gaps fs
        if (lane == 0) { double c = 0.0; for (int t = 0; t <= T; ++t) c += ws[L.fs + (long)t * kNDX]; sc[S_COST] = c; }
        __syncthreads();
        if (!feas) {
            double mx = 0.0;
            if (lane <= T) {
                double d[kNDX];
                const double *xa = ws + L.xs + (long)lane * kNX;
                const double *xb = lane == 0 ? a.x0 + b * kNX : ws + L.xnext + (long)(lane - 1) * kNX;
                state_diff<false>(xa, xb, d, nullptr);
                UNROLL_RBD for (int i = 0; i < kNDX; ++i) { ws[L.fs + (long)lane * kNDX + i] = d[i]; mx = fmax(mx, fabs(d[i])); }
            }
            if (lane == 0) s.flag = 0;
            __syncthreads();
            if (!(mx < 1e-16)) s.flag = 1;       // th_gaptol_
            __syncthreads();
            feas = s.flag == 0;
            __syncthreads();
            if (lane == 0) sc[S_FEAS] = feas ? 1.0 : 0.0;
        } else if (!wasfeas) {
            for (long i = lane; i < (long)(T + 1) * kNDX; i += kBwdThreads) ws[L.fs + i] = 0.0;
        } else {
            if (lane <= T) ws[L.fs + (long)lane * kNDX] = 0.0;   // the parked node costs
        }
        __syncthreads();
    }

    double xreg = sc[S_XREG];
    for (;;) {   // computeDirection with regularisation retries (solver-ddp.cpp solve())
        if (lane == 0) s.flag = 0;
        for (int e = lane; e < kNDX * kNDX; e += kBwdThreads) {
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
            for (int e = lane; e < kNDX * kNDX; e += kBwdThreads) s.W[(e / kNDX) * LD + e % kNDX] = ws[L.Lxx + (long)t * kNDX * kNDX + e];
            __syncthreads();
            // M1 = Fx^T V
            for (int e = lane; e < kNDX * kNDX; e += kBwdThreads) {
                const int i = e / kNDX, j = e % kNDX;
                double v;
                if (i < 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.A6[6 * c + i] * s.V[c * LD + j]; }
                else if (i < kNV) v = s.V[i * LD + j];
                else if (i < kNV + 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.B6[6 * c + (i - kNV)] * s.V[c * LD + j]; v = dt * v + s.V[i * LD + j]; }
                else v = dt * s.V[(i - kNV) * LD + j] + s.V[i * LD + j];
                s.M1[i * LD + j] = v;
            }
            // Qx = Lx + Fx^T Vx ; Qu = Lu + Fu^T Vx
            if (lane < kNDX) {
                const int i = lane;
                double v;
                if (i < 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.A6[6 * c + i] * s.Vx[c]; }
                else if (i < kNV) v = s.Vx[i];
                else if (i < kNV + 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.B6[6 * c + (i - kNV)] * s.Vx[c]; v = dt * v + s.Vx[i]; }
                else v = dt * s.Vx[i - kNV] + s.Vx[i];
                s.Qx[i] = ws[L.Lx + (long)t * kNDX + i] + v;
            }
            if (lane < kNV) {
                const int q = lane;
                double v;
                if (q < 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.B6[6 * c + q] * s.Vx[c]; } else v = s.Vx[q];
                s.Qu[q] = ws[L.Lu + (long)t * kNV + q] + dt2 * v + dt * s.Vx[kNV + q];
            }
            __syncthreads();
            // W = Lxx + M1 Fx ; Qxu = M1 Fu ; VFu = V Fu
            for (int e = lane; e < kNDX * kNDX; e += kBwdThreads) {
                const int i = e / kNDX, j = e % kNDX;
                double v;
                if (j < 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.M1[i * LD + c] * s.A6[6 * c + j]; }
                else if (j < kNV) v = s.M1[i * LD + j];
                else if (j < kNV + 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.M1[i * LD + c] * s.B6[6 * c + (j - kNV)]; v = dt * v + s.M1[i * LD + j]; }
                else v = dt * s.M1[i * LD + (j - kNV)] + s.M1[i * LD + j];
                s.W[i * LD + j] += v;
            }
            for (int e = lane; e < kNDX * kNV; e += kBwdThreads) {
                const int i = e / kNV, q = e % kNV;
                double v1, v2;
                if (q < 6) {
                    v1 = 0.0; v2 = 0.0;
                    UNROLL_RBD for (int c = 0; c < 6; ++c) { v1 += s.M1[i * LD + c] * s.B6[6 * c + q]; v2 += s.V[i * LD + c] * s.B6[6 * c + q]; }
                } else { v1 = s.M1[i * LD + q]; v2 = s.V[i * LD + q]; }
                s.Qxu[i * LDU + q] = dt2 * v1 + dt * s.M1[i * LD + kNV + q];
                s.VFu[i * LDU + q] = dt2 * v2 + dt * s.V[i * LD + kNV + q];
            }
            __syncthreads();
            // Quu = Luu + Fu^T (V Fu) + ureg I
            for (int e = lane; e < kNV * kNV; e += kBwdThreads) {
                const int p = e / kNV, q = e % kNV;
                double v;
                if (p < 6) { v = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) v += s.B6[6 * c + p] * s.VFu[c * LDU + q]; } else v = s.VFu[p * LDU + q];
                v = dt2 * v + dt * s.VFu[(kNV + p) * LDU + q];
                if (p == q) v += s.Luu[p] + xreg;
                s.Quu[p * LDU + q] = v;
            }
            __syncthreads();
            // Cholesky (lower, in place); a non-positive or NaN pivot fails the pass (Eigen::LLT info != Success)
            // (columns stay unscaled during the elimination -- one barrier per step -- and are divided by
            //  sqrt(pivot) in a single pass afterwards)
            for (int j = 0; j < kNV; ++j) {
                const double piv = s.Quu[j * LDU + j];
                if (!(piv > 0.0)) { if (lane == 0) s.flag = 1; }
                const double ip = 1.0 / piv;
                for (int e = lane; e < kNV * kNV; e += kBwdThreads) {
                    const int p = e / kNV, q = e % kNV;
                    if (q > j && p >= q) s.Quu[p * LDU + q] -= s.Quu[p * LDU + j] * s.Quu[q * LDU + j] * ip;
                }
                __syncthreads();
            }
            for (int e = lane; e < kNV * kNV; e += kBwdThreads) {
                const int p = e / kNV, q = e % kNV;
                if (p > q) s.Quu[p * LDU + q] /= sqrt(s.Quu[q * LDU + q]);
            }
            __syncthreads();
            if (lane < kNV) { const double d = sqrt(s.Quu[lane * LDU + lane]); s.Quu[lane * LDU + lane] = d; s.idg[lane] = 1.0 / d; }
            __syncthreads();
            // K = Quu^-1 Qxu^T: lane j < 36 solves for column j (18 unknowns in registers); lane 36: k = Quu^-1 Qu
            if (lane <= kNDX) {
                double y[kNV];
                UNROLL_RBD for (int p = 0; p < kNV; ++p) {
                    double v = lane < kNDX ? s.Qxu[lane * LDU + p] : s.Qu[p];
                    UNROLL_RBD for (int q = 0; q < p; ++q) v -= s.Quu[p * LDU + q] * y[q];
                    y[p] = v * s.idg[p];
                }
                UNROLL_RBD for (int p = kNV - 1; p >= 0; --p) {
                    double v = y[p];
                    UNROLL_RBD for (int q = p + 1; q < kNV; ++q) v -= s.Quu[q * LDU + p] * y[q];
                    y[p] = v * s.idg[p];
                }
                if (lane < kNDX) { UNROLL_RBD for (int p = 0; p < kNV; ++p) s.Kt[p * LD + lane] = y[p]; }
                else { UNROLL_RBD for (int p = 0; p < kNV; ++p) s.kf[p] = y[p]; }
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
            for (int e = lane; e < kNV * kNDX; e += kBwdThreads) ws[L.K + (long)t * kNV * kNDX + e] = s.Kt[(e / kNDX) * LD + e % kNDX];
            // Vx = Qx - K^T Qu ; Vxx = Qxx - Qxu K (into M1), then symmetrise + xreg
            if (lane < kNDX) { double v = s.Qx[lane]; UNROLL_RBD for (int p = 0; p < kNV; ++p) v -= s.Kt[p * LD + lane] * s.Qu[p]; s.Vx[lane] = v; }
            for (int e = lane; e < kNDX * kNDX; e += kBwdThreads) {
                const int i = e / kNDX, j = e % kNDX;
                double v = s.W[i * LD + j];
                UNROLL_RBD for (int p = 0; p < kNV; ++p) v -= s.Qxu[i * LDU + p] * s.Kt[p * LD + j];
                s.M1[i * LD + j] = v;
            }
            __syncthreads();
            bool bad = false;
            for (int e = lane; e < kNDX * kNDX; e += kBwdThreads) {
                const int i = e / kNDX, j = e % kNDX;
                const double v = 0.5 * (s.M1[i * LD + j] + s.M1[j * LD + i]) + (i == j ? xreg : 0.0);
                s.V[i * LD + j] = v;
                bad = bad || !(fabs(v) < INFINITY);
            }
            __syncthreads();
            if (!feas && lane < kNDX) { double acc = 0.0; for (int j = 0; j < kNDX; ++j) acc += s.V[lane * LD + j] * s.fs[j]; s.tmp[lane] = s.Vx[lane] + acc; }
            __syncthreads();
            if (!feas && lane < kNDX) s.Vx[lane] = s.tmp[lane];
            if (lane < kNDX) bad = bad || !(fabs(s.Vx[lane]) < INFINITY);   // raiseIfNaN on Vx / Vxx
            if (bad) s.flag = 1;
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
    double d1 = 0.0, d2 = 0.0, st = 0.0;
    __syncthreads();
    if (lane >= 64) return;
    for (int e = lane; e < T * kNV; e += 64) {
        const double qu = ws[L.Qu + e], kk = ws[L.kff + e];
        d1 += qu * kk; d2 -= kk * ws[L.Quuk + e]; st += qu * qu;
    }
    for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_down(d1, off); d2 += __shfl_down(d2, off); st += __shfl_down(st, off); }
    if (lane == 0) { sc[S_D1] = d1; sc[S_D2] = d2; sc[S_STOP] = st; }
}

// ---------------------------------------------------------------------------- forward ---
struct ForwardLds { RobotModelDev m; double dx[kNDX], u[kNV], x[kNX], xn[kNX]; double part[kLegs + 1][10 + 3 * kFrameSlots]; double bc[4]; };

__global__ __launch_bounds__(64) void ik_forward_kernel(const IkBatchArgs a) {
    __shared__ ForwardLds s;
    const long b = blockIdx.x;
    const int lane = threadIdx.x;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    double *sc = ws + L.scal;
    if (sc[S_DONE] != 0.0) return;
    const int T = a.T, nn = a.T + 1;
    {   // the robot model is read many times per node: stage it in LDS once
        const int *src = reinterpret_cast<const int *>(a.model);
        int *dst = reinterpret_cast<int *>(&s.m);
        for (int i = lane; i < (int)(sizeof(RobotModelDev) / sizeof(int)); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const RobotModelDev &m = s.m;
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
        if (lane < kNX) s.x[lane] = ws[L.xs_try + lane];
        __syncthreads();
        for (int t = 0; t <= T && !failed; ++t) {   // t == T: terminal node (cost only)
            const bool terminal = t == T;
            if (!terminal) {
                if (lane == 0) state_diff<false>(ws + L.xs + (long)t * kNX, s.x, s.dx, nullptr);
                __syncthreads();
                if (lane < kNV) {
                    const double *Kr = ws + L.K + (long)t * kNV * kNDX + (long)lane * kNDX;
                    double v = ws[L.us + (long)t * kNV + lane] - alpha * ws[L.kff + (long)t * kNV + lane];
                    UNROLL_RBD for (int j = 0; j < kNDX; ++j) v -= Kr[j] * s.dx[j];
                    s.u[lane] = v;
                    ws[L.us_try + (long)t * kNV + lane] = v;
                }
                __syncthreads();
            }
            // node evaluation spread over lanes: 0..3 legs, 4 base body, 5 state residual, 6 control cost + Euler step
            NodeTasks tk{a.tasks + (b * nn + t) * kNodeTaskDoubles};
            const double dtn = terminal ? 0.0 : a.dt[b * T + t];
            if (lane <= kLegs) {
                int fid[kFrameSlots];
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) fid[f] = tk.frame_w(f) != 0.0 ? tk.frame_id(f) : -1;
                PartSum ps;
                quad_part(m, s.x, fid, lane, ps);
                s.part[lane][0] = ps.mass;
                UNROLL_RBD for (int c = 0; c < 3; ++c) s.part[lane][1 + c] = ps.h1[c];
                UNROLL_RBD for (int c = 0; c < 6; ++c) s.part[lane][4 + c] = ps.hO[c];
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
                    UNROLL_RBD for (int c = 0; c < 3; ++c) s.part[lane][10 + 3 * f + c] = ps.fhit[f] ? ps.fx[f][c] : 0.0;
            } else if (lane == 5) {
                double acc = 0.0;
                if (tk.state_w() != 0.0) {
                    double rs[kNDX];
                    state_diff<false>(x_reg, s.x, rs, nullptr);
                    UNROLL_RBD for (int i = 0; i < kNDX; ++i) acc += state_w[i] * rs[i] * rs[i];
                }
                s.bc[2] = tk.state_w() * 0.5 * acc;
            } else if (lane == 6) {
                double acc = 0.0;
                if (!terminal) {
                    UNROLL_RBD for (int i = 0; i < kNV; ++i) acc += ctrl_w[i] * s.u[i] * s.u[i];
                    double xn[kNX];
                    euler_step<false>(s.x, s.u, dtn, xn, nullptr, nullptr);
                    bool bad = false;
                    UNROLL_RBD for (int i = 0; i < kNX; ++i) { ws[L.xs_try + (long)(t + 1) * kNX + i] = xn[i]; s.xn[i] = xn[i]; bad = bad || !(fabs(xn[i]) < INFINITY); }
                    s.bc[1] = bad ? 1.0 : 0.0;
                } else s.bc[1] = 0.0;
                s.bc[3] = tk.ctrl_w() * 0.5 * acc;
            }
            __syncthreads();
            if (lane == 0) {   // add the parts: CoM, centroidal momentum, residual costs
                double M = 0.0, h1[3] = {0, 0, 0}, hO[6] = {0, 0, 0, 0, 0, 0}, fx[kFrameSlots][3];
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) fx[f][0] = fx[f][1] = fx[f][2] = 0.0;
                UNROLL_RBD for (int pa = 0; pa <= kLegs; ++pa) {
                    M += s.part[pa][0];
                    UNROLL_RBD for (int c = 0; c < 3; ++c) h1[c] += s.part[pa][1 + c];
                    UNROLL_RBD for (int c = 0; c < 6; ++c) hO[c] += s.part[pa][4 + c];
                    UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
                        UNROLL_RBD for (int c = 0; c < 3; ++c) fx[f][c] += s.part[pa][10 + 3 * f + c];
                }
                double com[3], t3[3], c = 0.0, acc = 0.0;
                UNROLL_RBD for (int k = 0; k < 3; ++k) com[k] = h1[k] / M;
                cross3(com, hO, t3);
                UNROLL_RBD for (int k = 0; k < 3; ++k) {
                    const double rl = hO[k] - tk.mom_ref()[k], ra = hO[3 + k] - t3[k] - tk.mom_ref()[3 + k];
                    acc += rl * rl + ra * ra;
                }
                c += tk.mom_w() * 0.5 * acc;
                acc = 0.0;
                UNROLL_RBD for (int k = 0; k < 3; ++k) { const double r = com[k] - tk.com_ref()[k]; acc += r * r; }
                c += tk.com_w() * 0.5 * acc;
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) {
                    const double w = tk.frame_w(f);
                    acc = 0.0;
                    UNROLL_RBD for (int k = 0; k < 3; ++k) { const double r = w != 0.0 ? fx[f][k] - tk.frame_ref(f)[k] : 0.0; acc += r * r; }
                    c += w * 0.5 * acc;
                }
                c += s.bc[2] + s.bc[3];
                if (!terminal) c *= dtn;
                if (!(fabs(c) < INFINITY)) s.bc[1] = 1.0;
                s.bc[0] = c;
            }
            __syncthreads();
            if (lane < kNX && !terminal) s.x[lane] = s.xn[lane];
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
    Pass1 p1;
    const int fid[kFrameSlots] = {-1, -1, -1, -1};
    quad_pass1<false>(*model, x + b * kNX, fid, p1);
    for (int c = 0; c < 3; ++c) {
        out9[b * 9 + c] = p1.com[c];
        out9[b * 9 + 3 + c] = p1.hg[c] / p1.M;     // vcom
        out9[b * 9 + 6 + c] = p1.hg[3 + c];        // hg.angular
    }
}

__global__ void ik_com_mom_kernel(const RobotModelDev *model, const double *xs, double *com, double *mom, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pass1 p1;
    const int fid[kFrameSlots] = {-1, -1, -1, -1};
    quad_pass1<false>(*model, xs + i * kNX, fid, p1);
    for (int c = 0; c < 3; ++c) com[i * 3 + c] = p1.com[c];
    for (int c = 0; c < 6; ++c) mom[i * 6 + c] = p1.hg[c];
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
    hipLaunchKernelGGL(ik_calcdiff_kernel, dim3((unsigned)n), dim3(64), 0, st, a);
    return hipGetLastError();
}
hipError_t ik_launch_backward(const IkBatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ik_backward_kernel, dim3(a.B), dim3(kBwdThreads), 0, st, a);
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

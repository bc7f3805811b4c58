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
//   ik_backward_kernel  one WAVE per problem, matrix rows in registers (lane r = row r of V, G, Q_xx), exploiting
//                       F_x = [[A, dt B],[0, I]], F_u = dt F_x[:, v] (A, B identity except a 6x6 free-flyer block):
//                       G = F_x^T V F_x via one LDS transposition, Cholesky and the gain solves in registers
//                       over v_readlane, V_xx = Q_xx - Q_xu K against broadcast LDS reads; regularisation
//                       retries inside the kernel (details above the kernel).
//   ik_forward_kernel   one WAVE per problem: line search 2^-k, k = 0..9 -- lanes 0..17 apply the
//                       feedback u = u - a k - K dx, the node evaluation is spread over lanes (legs, base,
//                       state cost, control cost + Euler step); acceptance, regularisation update and
//                       stopping test as crocoddyl 1.9.0 solver-ddp.cpp.
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

__global__ __launch_bounds__(64, 2) void ik_calcdiff_kernel(const IkBatchArgs a) {
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
// One wave per problem, no cross-wave barrier.  Lane r < 36 owns ROW r of every 36x36 matrix of the
// Riccati step in registers; LDS (14.6 KB per wave) is only the exchange medium: one transposition per
// node, the 18x18 Cholesky factor, the gain matrix (read back by broadcast) and a few vectors.
//
// With the integrator's F_u = dt * F_x[:, v-columns] (F_x = [[A, dt B],[0, I]], F_u = [[dt^2 B],[dt I]]),
// everything the step needs is a slice of  G = F_x^T V F_x :
//     Q_xx = L_xx + G,   Q_xu = dt G[:, v],   Q_uu = L_uu + dt^2 G[v, v],   Q_u = L_u + dt (F_x^T V_x)[v]
// and since V is symmetric, lane r computes column r of N = F_x^T V from its own row of V; one LDS
// transposition later it holds row r of N and finishes row r of G = N F_x, again lane-locally.
constexpr int LD = kNDX + 1;   // odd leading dimension: rows and columns of 64-bit words are both conflict-free
constexpr int LDU = kNV + 1;

// value of v in lane `src` (compile-time constant), wave-uniform: two v_readlane_b32
__device__ __forceinline__ double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// 1 / b to ~1 ulp: v_rcp_f64 + two Newton steps
__device__ __forceinline__ double rcp64(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    return fma(fma(-b, r, 1.0), r, r);
}

typedef double double2_t __attribute__((ext_vector_type(2)));
// byte offset of a __shared__ object inside the workgroup's LDS (low half of its flat address)
__device__ __forceinline__ unsigned lds_offset(const void *p) { return (unsigned)(unsigned long long)p; }
// Two consecutive 18-double columns (16-byte aligned) from LDS with all 18 ds_read_b128 in flight at once.
// The compiler's scheduler keeps every LDS read glued to its first use (one exposed latency per read);
// issuing the batch from one asm block costs one latency per batch instead.
__device__ __forceinline__ void lds_read_2x18(unsigned addr, double2_t (&a)[9], double2_t (&b)[9]) {
    asm volatile(
        "ds_read_b128 %0, %18\n\tds_read_b128 %1, %18 offset:16\n\tds_read_b128 %2, %18 offset:32\n\t"
        "ds_read_b128 %3, %18 offset:48\n\tds_read_b128 %4, %18 offset:64\n\tds_read_b128 %5, %18 offset:80\n\t"
        "ds_read_b128 %6, %18 offset:96\n\tds_read_b128 %7, %18 offset:112\n\tds_read_b128 %8, %18 offset:128\n\t"
        "ds_read_b128 %9, %18 offset:144\n\tds_read_b128 %10, %18 offset:160\n\tds_read_b128 %11, %18 offset:176\n\t"
        "ds_read_b128 %12, %18 offset:192\n\tds_read_b128 %13, %18 offset:208\n\tds_read_b128 %14, %18 offset:224\n\t"
        "ds_read_b128 %15, %18 offset:240\n\tds_read_b128 %16, %18 offset:256\n\tds_read_b128 %17, %18 offset:272\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]), "=&v"(a[8]),
          "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7]), "=&v"(b[8])
        : "v"(addr)
        : "memory");
}

// 36 consecutive doubles (8-byte aligned) with the 18 ds_read2_b64 in flight together
__device__ __forceinline__ void lds_read_row36(unsigned addr, double (&x)[kNDX]) {
    double2_t t[18];
    asm volatile("ds_read2_b64 %0, %18 offset0:0 offset1:1\n\tds_read2_b64 %1, %18 offset0:2 offset1:3\n\tds_read2_b64 %2, %18 offset0:4 offset1:5\n\tds_read2_b64 %3, %18 offset0:6 offset1:7\n\tds_read2_b64 %4, %18 offset0:8 offset1:9\n\tds_read2_b64 %5, %18 offset0:10 offset1:11\n\tds_read2_b64 %6, %18 offset0:12 offset1:13\n\tds_read2_b64 %7, %18 offset0:14 offset1:15\n\tds_read2_b64 %8, %18 offset0:16 offset1:17\n\tds_read2_b64 %9, %18 offset0:18 offset1:19\n\tds_read2_b64 %10, %18 offset0:20 offset1:21\n\tds_read2_b64 %11, %18 offset0:22 offset1:23\n\tds_read2_b64 %12, %18 offset0:24 offset1:25\n\tds_read2_b64 %13, %18 offset0:26 offset1:27\n\tds_read2_b64 %14, %18 offset0:28 offset1:29\n\tds_read2_b64 %15, %18 offset0:30 offset1:31\n\tds_read2_b64 %16, %18 offset0:32 offset1:33\n\tds_read2_b64 %17, %18 offset0:34 offset1:35\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15]), "=&v"(t[16]), "=&v"(t[17])
                 : "v"(addr) : "memory");
    UNROLL_RBD for (int i = 0; i < 18; ++i) { x[2 * i] = t[i].x; x[2 * i + 1] = t[i].y; }
}
// 36 doubles at stride LD (column r of the row-major staging matrix): two batches of 18 ds_read_b64
__device__ __forceinline__ void lds_read_col36(unsigned addr, double (&x)[kNDX]) {
    asm volatile("ds_read_b64 %0, %18 offset:0\n\tds_read_b64 %1, %18 offset:296\n\tds_read_b64 %2, %18 offset:592\n\tds_read_b64 %3, %18 offset:888\n\tds_read_b64 %4, %18 offset:1184\n\tds_read_b64 %5, %18 offset:1480\n\tds_read_b64 %6, %18 offset:1776\n\tds_read_b64 %7, %18 offset:2072\n\tds_read_b64 %8, %18 offset:2368\n\tds_read_b64 %9, %18 offset:2664\n\tds_read_b64 %10, %18 offset:2960\n\tds_read_b64 %11, %18 offset:3256\n\tds_read_b64 %12, %18 offset:3552\n\tds_read_b64 %13, %18 offset:3848\n\tds_read_b64 %14, %18 offset:4144\n\tds_read_b64 %15, %18 offset:4440\n\tds_read_b64 %16, %18 offset:4736\n\tds_read_b64 %17, %18 offset:5032\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17])
                 : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %18 offset:5328\n\tds_read_b64 %1, %18 offset:5624\n\tds_read_b64 %2, %18 offset:5920\n\tds_read_b64 %3, %18 offset:6216\n\tds_read_b64 %4, %18 offset:6512\n\tds_read_b64 %5, %18 offset:6808\n\tds_read_b64 %6, %18 offset:7104\n\tds_read_b64 %7, %18 offset:7400\n\tds_read_b64 %8, %18 offset:7696\n\tds_read_b64 %9, %18 offset:7992\n\tds_read_b64 %10, %18 offset:8288\n\tds_read_b64 %11, %18 offset:8584\n\tds_read_b64 %12, %18 offset:8880\n\tds_read_b64 %13, %18 offset:9176\n\tds_read_b64 %14, %18 offset:9472\n\tds_read_b64 %15, %18 offset:9768\n\tds_read_b64 %16, %18 offset:10064\n\tds_read_b64 %17, %18 offset:10360\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[18]), "=&v"(x[19]), "=&v"(x[20]), "=&v"(x[21]), "=&v"(x[22]), "=&v"(x[23]), "=&v"(x[24]), "=&v"(x[25]), "=&v"(x[26]), "=&v"(x[27]), "=&v"(x[28]), "=&v"(x[29]), "=&v"(x[30]), "=&v"(x[31]), "=&v"(x[32]), "=&v"(x[33]), "=&v"(x[34]), "=&v"(x[35])
                 : "v"(addr) : "memory");
}

struct alignas(16) BackwardLds {
    double N[kNDX * LD];       // row-major staging: N = F_x^T V for the transposition, later Q_xx -> V_xx rows
    double Kt[kNDX * kNV];     // K^T: column j of K contiguous at Kt[18 j], read back by broadcast
    double A6[36], B6[36];
    double Vx[kNDX], fs[kNDX];
};

// x <- F_x^T x for a 36-vector held by one lane, in place: only x[0..5] feed more than one output.
// A6 / B6 (row-major 6x6) are wave-uniform and come from LDS in one batch each.
__device__ __forceinline__ void apply_FxT(double (&x)[kNDX], unsigned a6_addr, unsigned b6_addr, double dt) {
    double t6[6], blk[36];
    UNROLL_RBD for (int c = 0; c < 6; ++c) t6[c] = x[c];
    lds_read_row36(a6_addr, blk);
    UNROLL_RBD for (int j = 0; j < 6; ++j) {
        double v = 0.0;
        UNROLL_RBD for (int c = 0; c < 6; ++c) v += blk[6 * c + j] * t6[c];
        x[j] = v;
    }
    lds_read_row36(b6_addr, blk);
    UNROLL_RBD for (int k = 0; k < 6; ++k) {
        double v = 0.0;
        UNROLL_RBD for (int c = 0; c < 6; ++c) v += blk[6 * c + k] * t6[c];
        x[kNV + k] = dt * v + x[kNV + k];
    }
    UNROLL_RBD for (int k = 6; k < kNV; ++k) x[kNV + k] = dt * x[k] + x[kNV + k];
}

__global__ __launch_bounds__(64, 2) void ik_backward_kernel(const IkBatchArgs a) {
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
            feas = !__any(!(mx < 1e-16));       // th_gaptol_
            if (lane == 0) sc[S_FEAS] = feas ? 1.0 : 0.0;
        } else if (!wasfeas) {
            for (long i = lane; i < (long)(T + 1) * kNDX; i += 64) ws[L.fs + i] = 0.0;
        } else {
            if (lane <= T) ws[L.fs + (long)lane * kNDX] = 0.0;   // the parked node costs
        }
        __syncthreads();
    }

    const bool row = lane < kNDX;                  // owns row `lane` of the 36x36 matrices
    const int r = row ? lane : 0;
    const bool ul = lane >= kNV && lane < kNDX;    // owns control q = lane - 18 (rows 18..35 are the v-rows)
    const int uq = ul ? lane - kNV : 0;
    const unsigned row_addr = lds_offset(s.N + r * LD), col_addr = lds_offset(s.N + r);
    const unsigned a6_addr = lds_offset(s.A6), b6_addr = lds_offset(s.B6), fs_addr = lds_offset(s.fs);
    double xreg = sc[S_XREG];
    double d1, d2, st;
#ifdef BWD_PROFILE
    long long pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#define PSTAMP(k) { const long long now_ = __builtin_readcyclecounter(); pc[k] += now_ - pt0; pt0 = now_; }
// stamp that cannot be passed by the computation of x (nor x's consumers hoisted above it)
#define PSTAMPV(k, x) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x) :: "memory"); PSTAMP(k) asm volatile("" : "+v"(x) :: "memory"); }
#else
#define PSTAMP(k)
#define PSTAMPV(k, x)
#endif
    for (;;) {   // computeDirection with regularisation retries (solver-ddp.cpp solve())
        bool bad = false;
        d1 = 0.0; d2 = 0.0; st = 0.0;
        double m[kNDX];   // the one 36-wide register row: V -> N -> G -> Q_xx -> V_xx -> V
        {
            const double *LT = ws + L.Lxx + (long)T * kNDX * kNDX + (long)r * kNDX;
            UNROLL_RBD for (int j = 0; j < kNDX; ++j) m[j] = LT[j] + (j == r ? xreg : 0.0);
        }
        double vx = row ? ws[L.Lx + (long)T * kNDX + r] : 0.0;
        if (!feas) {
            if (row) s.fs[r] = ws[L.fs + (long)T * kNDX + r];
            __syncthreads();
            double acc = 0.0, fsv[kNDX];
            lds_read_row36(fs_addr, fsv);
            UNROLL_RBD for (int j = 0; j < kNDX; ++j) acc += m[j] * fsv[j];
            vx += acc;
            __syncthreads();
        }

        for (int t = T - 1; t >= 0; --t) {
            const double dt = a.dt[b * T + t];
#ifdef BWD_PROFILE
            pt0 = __builtin_readcyclecounter();
#endif
            if (lane < 36) { s.A6[lane] = ws[L.A6 + (long)t * 36 + lane]; s.B6[lane] = ws[L.B6 + (long)t * 36 + lane]; }
            if (row) { s.Vx[r] = vx; s.fs[r] = ws[L.fs + (long)t * kNDX + r]; }
            __syncthreads();
            PSTAMPV(0, m[0])
            apply_FxT(m, a6_addr, b6_addr, dt);                  // column r of N = F_x^T V
            if (row) { UNROLL_RBD for (int i = 0; i < kNDX; ++i) s.N[i * LD + r] = m[i]; }
            __syncthreads();
            PSTAMPV(1, m[0])
            lds_read_row36(row_addr, m);                         // row r of N
            apply_FxT(m, a6_addr, b6_addr, dt);                  // row r of G = N F_x
            PSTAMPV(2, m[35])
            // Q_x = L_x + F_x^T V_x ;  Q_u = L_u + dt (F_x^T V_x)[v]
            double fvx;
            if (r < 6) { fvx = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) fvx += s.A6[6 * c + r] * s.Vx[c]; }
            else if (r < kNV) fvx = s.Vx[r];
            else if (r < kNV + 6) { fvx = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) fvx += s.B6[6 * c + (r - kNV)] * s.Vx[c]; fvx = dt * fvx + s.Vx[r]; }
            else fvx = dt * s.Vx[r - kNV] + s.Vx[r];
            const double qx = (row ? ws[L.Lx + (long)t * kNDX + r] : 0.0) + fvx;
            const double qu = ul ? ws[L.Lu + (long)t * kNV + uq] + dt * fvx : 0.0;   // Q_u[q] on lane 18 + q
            // Q_xu row; Q_uu row p on lane 18 + p (without its L_uu + reg diagonal term, kept in dgv); Q_xx row -> LDS
            double qxu[kNV], al[kNV];
            UNROLL_RBD for (int q = 0; q < kNV; ++q) { qxu[q] = dt * m[kNV + q]; al[q] = dt * qxu[q]; }
            const double dgv = ul ? ws[L.Luu + (long)t * kNV + uq] + xreg : 0.0;
            {
                const double *Lr = ws + L.Lxx + (long)t * kNDX * kNDX + (long)r * kNDX;
                if (row) { UNROLL_RBD for (int j = 0; j < kNDX; ++j) s.N[r * LD + j] = m[j] + Lr[j]; }
            }
            PSTAMPV(3, al[17])
            // Cholesky Q_uu = L L^T entirely in registers: lane 18 + p owns row p, the column entries and pivots
            // every lane needs travel by v_readlane (wave-uniform, no LDS, no waiting).  A non-positive or NaN
            // pivot fails the pass (Eigen::LLT info != Success).
            double idg[kNV];
            UNROLL_RBD for (int j = 0; j < kNV; ++j) {
                const double piv = lane_value(al[j], kNV + j) + lane_value(dgv, kNV + j);
                if (!(piv > 0.0)) bad = true;
                const double f = al[j] * rcp64(piv);
                UNROLL_RBD for (int q = j + 1; q < kNV; ++q) al[q] -= f * lane_value(al[j], kNV + q);
                idg[j] = rcp64(sqrt(piv));
                al[j] *= idg[j];                        // L[p][j] for the rows p > j
            }
            PSTAMPV(4, idg[17])
            // K = Quu^-1 Qxu^T: lane j < 36 solves for column j (18 unknowns in registers); lane 36: k = Quu^-1 Qu
            double y[kNV], quv[kNV];
            UNROLL_RBD for (int p = 0; p < kNV; ++p) quv[p] = lane_value(qu, kNV + p);
            UNROLL_RBD for (int p = 0; p < kNV; ++p) {
                double w = lane < kNDX ? qxu[p] : quv[p];
                UNROLL_RBD for (int q = 0; q < p; ++q) w -= lane_value(al[q], kNV + p) * y[q];
                y[p] = w * idg[p];
            }
            // expectedImprovement / stoppingCriteria ingredients (lane 36): d2 = -k.Quu k = -|L^T k|^2 = -|L^-1 Qu|^2
            UNROLL_RBD for (int p = 0; p < kNV; ++p) d2 -= y[p] * y[p];
            UNROLL_RBD for (int p = kNV - 1; p >= 0; --p) {
                double w = y[p];
                UNROLL_RBD for (int q = p + 1; q < kNV; ++q) w -= lane_value(al[p], kNV + q) * y[q];
                y[p] = w * idg[p];
            }
            PSTAMPV(5, y[0])
            UNROLL_RBD for (int p = 0; p < kNV; ++p) { d1 += quv[p] * y[p]; st += quv[p] * quv[p]; }   // d1 = Qu.k, stop = |Qu|^2
            double *Ks = s.Kt;
            if (row) {
                UNROLL_RBD for (int p = 0; p < kNV; ++p) { Ks[r * kNV + p] = y[p]; ws[L.K + (long)t * kNV * kNDX + (long)p * kNDX + r] = y[p]; }
            } else if (lane == kNDX) {
                UNROLL_RBD for (int p = 0; p < kNV; ++p) ws[L.kff + (long)t * kNV + p] = y[p];
            }
            // V_x = Q_x - K^T Q_u
            {
                double w = qx;
                UNROLL_RBD for (int p = 0; p < kNV; ++p) w -= y[p] * quv[p];
                vx = w;
            }
            __syncthreads();
            PSTAMPV(6, vx)
            // V_xx = Q_xx - Q_xu K, row r, in its LDS row: the column loop is a real loop (36 x 18 fused multiply-adds
            // against broadcast reads of K^T; unrolled over all j the scheduler hoists every read and spills)
            {
                const unsigned kaddr = lds_offset(Ks);
                for (int j0 = 0; j0 < kNDX; j0 += 2) {      // two columns per trip, their 18 K^T reads in flight together
                    double2_t k0[9], k1[9];
                    lds_read_2x18(kaddr + (unsigned)j0 * (kNV * 8), k0, k1);
                    double w0 = s.N[r * LD + j0], w1 = s.N[r * LD + j0 + 1];
                    UNROLL_RBD for (int p = 0; p < 9; ++p) {
                        w0 -= qxu[2 * p] * k0[p].x; w1 -= qxu[2 * p] * k1[p].x;
                        w0 -= qxu[2 * p + 1] * k0[p].y; w1 -= qxu[2 * p + 1] * k1[p].y;
                    }
                    if (row) { s.N[r * LD + j0] = w0; s.N[r * LD + j0 + 1] = w1; }
                }
            }
            PSTAMPV(7, vx)
            // V = (V_xx + V_xx^T)/2 + xreg I: own row and own column of the staged V_xx (xreg added to the staged diagonal)
            if (row) s.N[r * LD + r] += xreg;
            __syncthreads();
            lds_read_row36(row_addr, m);
            {
                double col[kNDX];
                lds_read_col36(col_addr, col);
                UNROLL_RBD for (int j = 0; j < kNDX; ++j) {
                    m[j] = 0.5 * (m[j] + col[j]);
                    bad = bad || !(fabs(m[j]) < INFINITY);
                }
            }
            if (!feas) {
                double acc = 0.0, fsv[kNDX];
                lds_read_row36(fs_addr, fsv);
                UNROLL_RBD for (int j = 0; j < kNDX; ++j) acc += m[j] * fsv[j];
                vx += acc;
            }
            bad = bad || !(fabs(vx) < INFINITY);       // raiseIfNaN on Vx / Vxx
            bad = __any(bad && (row || lane < kNV));
            __syncthreads();
            PSTAMPV(8, vx)
            if (bad) break;
        }
        if (!bad) break;
        // increaseRegularization; give up at reg_max (solve() returns false)
        xreg = fmin(xreg * 10.0, 1e9);
        if (lane == 0) { sc[S_XREG] = xreg; sc[S_RECALC] = 0.0; }
        if (xreg == 1e9) {
            if (lane == 0) { sc[S_DONE] = 1.0; sc[S_STATUS] = 2.0; atomicSub(a.active, 1); }
            return;
        }
    }
#ifdef BWD_PROFILE
    if (lane == 0) { for (int k = 0; k < 9; ++k) ws[L.Quuk + k] = (double)pc[k]; }   // the Quuk slot is unused by the solver
#endif
    if (lane == kNDX) { sc[S_D1] = d1; sc[S_D2] = d2; sc[S_STOP] = st; }   // lane 36 solved for the feed-forward terms
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

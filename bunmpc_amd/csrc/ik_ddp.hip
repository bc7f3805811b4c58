// Batched whole-body inverse-kinematics DDP for gfx950 (MI355X).
//
// Restates ik::InverseKinematics::optimize (ISL/src/ik/inverse_kinematics.cpp:54-71): a crocoddyl
// ShootingProblem of IntegratedActionModelEuler nodes around the "kinematic" differential model
// (ISL/src/ik/action_model.cpp:43-94: xout = u, Fx = 0, Fu = I) with the residual costs of
// ISL/src/ik/{com_tasks,end_effector_tasks,regularization_costs}.cpp, solved by
// crocoddyl::SolverDDP::solve() with all defaults.  crocoddyl 1.9.0 / pinocchio 2.6.9 are third
// party and absent: semantics follow oracle/ik_ddp_np.py (PARITY UNPINNED).
//
// MI355X organisation (nothing like crocoddyl's object graph).  All per-problem state lives in one contiguous HBM workspace
// (IkLayout); the DEVICE PIECES below are shared by every mapping, so a problem's result never depends on how it was scheduled
// (the tests compare bit for bit):
//   state_node          the scalar chains of a node's derivative pass (state residual with its Jlog6 block, Euler step with its
//                       Jintegrate blocks, node cost parts): one LANE per (problem, node)  [ik_state_kernel]
//   calc_walk / calc_columns / calc_assemble
//                       one robot walk per node (lane = velocity column, part sums through LDS), the residual Jacobian's
//                       columns, the Gauss-Newton L_x / L_xx (3 x 3 tiles of v_mfma_f64_16x16x4)
//                       [ik_calcdiff_kernel: two waves per node pair; ik_calcdiff1_kernel: ONE wave per pair when a launch has
//                       more pairs than the chip holds]
//   backward_main_wave (+ backward_gains_wave)
//                       one WAVE per problem, matrix rows in registers (lane r = row r of V, G, Q_xx), exploiting
//                       F_x = [[A, dt B],[0, I]], F_u = dt F_x[:, v-columns]: G = F_x^T V F_x via one LDS transposition,
//                       Cholesky in registers over v_readlane with the forward substitutions riding along, V_xx = Q_xx - Y^T Y
//                       on the matrix pipe, regularisation retries inside; with few problems a second wave computes and stores
//                       the gains one node behind  [ik_backward_kernel<1|2>]
//   forward_body<NW, FUSED, ROLES>
//                       line search 2^-k + rollout: FOUR problems per wave, or four step lengths of one problem on two / three
//                       waves, each wave a separately compiled instantiation of its role (chain / robot walks / state
//                       residual); all ten step lengths on three workgroups for the problems flagged as needing them;
//                       acceptance, regularisation update and stopping test as crocoddyl 1.9.0 solver-ddp.cpp
//                       [ik_forward_kernel<1|2|3>]
// Two ways of running them:
//   * LOCK-STEP: four launches per DDP iteration over the problems still iterating (an active list the forward pass rebuilds);
//     the host loops and stops when the device-side counter reaches zero (bunmpc_ik_capi.hip::run_ddp);
//   * FUSED: ik_fused_kernel, one persistent four-wave workgroup per problem running whole DDP iterations on chip with no host
//     look (derivative pass pipelined behind the Riccati pass over shared barrier "ticks", then the line search), fed by the
//     EXPRESS LANE: ik_select_kernel takes the problems farthest from converging off a fast-converging batch after its third
//     iteration, and they run to the end on a side stream while the batch goes on without them.
#include "ik_types.h"
#include "rbd_quad.h"
#include "lds_batch.h"

namespace bunmpc {
namespace {

using namespace rbd;

// -DBWD_PROFILE: cycle stamps inside the kernels (tools/bwd_profile.py reads them back); PSTAMPV cannot be passed
// by the computation of x
#ifdef BWD_PROFILE
#define PSTAMP(k) { const long long now_ = __builtin_readcyclecounter(); pc[k] += now_ - pt0; pt0 = now_; }
#define PSTAMPV(k, x) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x) :: "memory"); PSTAMP(k) asm volatile("" : "+v"(x) :: "memory"); }
#elif defined(BWD_MARK)
#define PSTAMP(k)
#define PSTAMPV(k, x) asm volatile("; BWDMARK " #k ::: "memory");
#else
#define PSTAMP(k)
#define PSTAMPV(k, x)
#endif

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
    state_integrate_q(x, dx, xnext);
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

// index checks of the list code (ik_types.h::IkIndexError): the first failure is recorded, the access dropped
__device__ __forceinline__ void index_error(const IkBatchArgs &a, int code, int value) {
    if (a.err && atomicCAS(a.err, 0, code) == 0) a.err[1] = value;
}
// problem handled by launch slot `slot` of this DDP iteration: through the active list when there is one (-1: past its end)
__device__ __forceinline__ long slot_problem(const IkBatchArgs &a, long slot) {
    if (!a.list) return slot < a.B ? slot : -1;
    const int cur = a.iter & 1, n = a.count[cur];
    if ((unsigned)n > (unsigned)a.B) { index_error(a, IK_ERR_LIST_COUNT, n); return -1; }
    if (slot >= n) return -1;
    const int p = a.list[(long)cur * a.B + slot];
    if ((unsigned)p >= (unsigned)a.B) { index_error(a, IK_ERR_LIST_ENTRY, p); return -1; }
    return p;
}

// A workgroup of ONE wave (the backward pass; the one-wave derivative pass): its LDS accesses execute in program order, so what a barrier has to provide
// is only that the compiler keeps them in that order and that earlier LDS operations have completed.  __syncthreads() would
// also wait for every outstanding GLOBAL load (s_waitcnt vmcnt(0)) -- here that is the prefetched L_xx row of the node, which
// is not needed before the elimination and should keep flying across the exchanges in between.
// Sum of one value per node, taken in the order SolverDDP::calcDiff adds the node costs (node 0 first): lane t holds node t's
// value (T + 1 <= 64), every lane ends up with the sum.  v_readlane hands the values to the additions one by one.
template <class F>
__device__ __forceinline__ double sum_nodes_in_turn(F node_value, int T) {      // node_value(t) for t <= T, whatever for t > T (not added)
    const int lane = threadIdx.x & 63;
    double c = 0.0;
    for (int base = 0; base <= T; base += 64) {      // (64 nodes at a time: their loads go out side by side, one node per lane)
        const int node = base + lane;
        const double mine = node <= T ? node_value(node) : 0.0;
        const int n = T - base < 63 ? T - base : 63;
        for (int t = 0; t <= n; ++t)
            c += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), t), __builtin_amdgcn_readlane(__double2loint(mine), t));
    }
    return c;
}
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------- init ---
// The robot model into LDS, by all threads of the workgroup: eight words per thread in flight (one by one, each load was waited
// for before its LDS write: 6 .. 18 round trips to L2 in front of a kernel's first node)
template <int THREADS>
__device__ __forceinline__ void stage_model(const RobotModelDev *model, RobotModelDev *lds) {
    constexpr int kWords = (int)(sizeof(RobotModelDev) / sizeof(int));
    const int *src = reinterpret_cast<const int *>(model);
    int *dst = reinterpret_cast<int *>(lds);
    for (int i0 = threadIdx.x; i0 < kWords; i0 += 8 * THREADS) {
        int v[8];
        UNROLL_RBD for (int k = 0; k < 8; ++k) { const int i = i0 + k * THREADS; v[k] = src[i < kWords ? i : kWords - 1]; }
        UNROLL_RBD for (int k = 0; k < 8; ++k) { const int i = i0 + k * THREADS; if (i < kWords) dst[i] = v[k]; }
    }
}

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
    s[S_DONE] = 0; s[S_ITERS] = 0; s[S_RECALC] = 1; s[S_STATUS] = 0; s[S_WIDE] = 0; s[S_WIDENOW] = 0;
    ws[L.arrive] = 0.0;          // (read as an unsigned counter)
    for (int k = 0; k < 12; ++k) ws[L.Qu + k] = 0.0;     // the fused kernel's telemetry (the Qu slot is unused by the solver)
    if (a.list) a.list[b] = (int)b;
    if (b == 0) { *a.active = a.B; if (a.count) { a.count[0] = a.B; a.count[1] = 0; a.wcount[0] = 0; a.wcount[1] = 0; a.err[0] = 0; a.err[1] = 0; a.near[0] = 0; a.near[1] = 0;
                                               for (int k = 0; k < 4; ++k) a.xmeta[k] = 0; } }
}

// --------------------------------------------------------------------------- calcDiff ---
// Residual rows of one node, stored by COLUMN of the state tangent: Jt[j] = [dh_g/dx (6) | dCoM/dq (3) |
// frame translations (4 x 3) | pad]; the velocity half (j >= 18) carries A_g in the momentum rows only.
constexpr int kRes = 6 + 3 + 3 * kFrameSlots;   // 21 residual rows
constexpr int kResLd = kRes + 1;                // 22 doubles = 11 x 16 bytes per column

typedef double mfma_acc_t __attribute__((ext_vector_type(4)));
#define CALC_STR2(x) #x
#define CALC_STR(x) CALC_STR2(x)
#define CALC_RD(n, I, ks) "ds_read_b64 %" #n ", %18 offset:" CALC_STR(((16 * I * 22 + 4 * ks) * 8)) "\n"
// o[6 I + ks] = Jt[16 I + (l & 15)][4 ks + (l >> 4)] (rows of kResLd = 22 doubles): the MFMA operand elements of a lane
__device__ __forceinline__ void lds_read_calc_operand(unsigned base, double (&o)[18]) {
    static_assert(kResLd == 22, "offsets below are written for rows of 22 doubles");
    asm volatile(CALC_RD(0, 0, 0) CALC_RD(1, 0, 1) CALC_RD(2, 0, 2) CALC_RD(3, 0, 3) CALC_RD(4, 0, 4) CALC_RD(5, 0, 5)
                 CALC_RD(6, 1, 0) CALC_RD(7, 1, 1) CALC_RD(8, 1, 2) CALC_RD(9, 1, 3) CALC_RD(10, 1, 4) CALC_RD(11, 1, 5)
                 CALC_RD(12, 2, 0) CALC_RD(13, 2, 1) CALC_RD(14, 2, 2) CALC_RD(15, 2, 3) CALC_RD(16, 2, 4) CALC_RD(17, 2, 5)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]),
                   "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17])
                 : "v"(base) : "memory");
}

constexpr int kParts = kLegs + 1;        // legs 0..3, base body 4
constexpr int kPartDoubles = 16;         // Comp (m, h1[3], I[6]) + momentum about the origin (6)

struct alignas(16) CalcNode {         // per node of the workgroup
    double Jt[kNDX][kResLd];
    double JlT[6][6];                 // Jlog6 block of the state residual Jacobian, transposed: JlT[i][k] = Jl[6 k + i]
    double parts[kParts][kPartDoubles];
    double res[kResLd];               // residuals [momentum 6 | CoM 3 | frames 12 | pad]
    double fx[kFrameSlots][3];        // task-frame positions (written by the part that carries the frame)
    double rs[kNDX];                  // state residual
    double cost_kin, cost_sc;         // node cost: momentum + CoM + frames | state + control
    double x[kNX + 1], u[kNV];
    // The node's task block and its time step (terminal: 0), staged with the state in ONE coalesced load.  Read field by field
    // from global memory -- `frame_w(f) != 0 ? frame_id(f) : -1`, the references, the weights: some sixty loads per node pair, each
    // behind a condition or a call, each waited for on its own (`global_load ... s_waitcnt vmcnt(0)` sixty times in the ISA) -- they
    // were what ik_calcdiff1 spent 61 % of its wave cycles waiting for.
    double tk[kNodeTaskDoubles + 1];
};
// the one-step form (the fused kernel's producers: the two-step form there trips a back-end error of this hipcc, "Illegal instruction
// detected: V_CMP_NE_U32_e32 0, $src_shared_base", where the LDS node is reached through a generic reference)
__device__ __forceinline__ void calc_stage_simple(const IkBatchArgs &a, long b, int t, const double *ws, const IkLayout &L, CalcNode &q, int l) {
    const int nn = a.T + 1;
    if (l >= 0 && l < kNX) q.x[l] = ws[L.xs + (long)t * kNX + l];
    if (l >= 0 && l < kNV) q.u[l] = t == a.T ? 0.0 : ws[L.us + (long)t * kNV + l];
    if (l >= 0 && l < kNodeTaskDoubles) q.tk[l] = a.tasks[(b * nn + t) * kNodeTaskDoubles + l];
    if (l == kNodeTaskDoubles) q.tk[l] = t == a.T ? 0.0 : a.dt[b * a.T + t];
}
// Stage node t of problem b into q: lane l < 38 takes one element of x, u and the task block each.  In two steps -- every load of
// the node (of BOTH nodes of a pair, in the callers) is requested before the first LDS store waits for one: as `q.x[l] = ws[...]`
// behind a lane condition each of the eight was a load, a wait and a store of its own.  (Every lane loads an existing element.)
struct StagedNode { double x, u, tk, dt; };
__device__ __forceinline__ StagedNode calc_stage_load(const IkBatchArgs &a, long b, int t, const double *ws, const IkLayout &L, int l) {
    const int nn = a.T + 1, tu = t < a.T ? t : a.T - 1;
    StagedNode v;
    v.x = ws[L.xs + (long)t * kNX + (l < kNX ? l : 0)];
    v.u = ws[L.us + (long)tu * kNV + (l < kNV ? l : 0)];
    v.tk = a.tasks[(b * nn + t) * kNodeTaskDoubles + (l < kNodeTaskDoubles ? l : 0)];
    v.dt = a.dt[b * a.T + tu];
    return v;
}
__device__ __forceinline__ void calc_stage_store(const IkBatchArgs &a, int t, CalcNode &q, int l, const StagedNode &v) {
    if (l < kNX) q.x[l] = v.x;
    if (l < kNV) q.u[l] = t == a.T ? 0.0 : v.u;
    if (l < kNodeTaskDoubles) q.tk[l] = v.tk;
    if (l == kNodeTaskDoubles) q.tk[l] = t == a.T ? 0.0 : v.dt;
}
constexpr int kCalcNodes = 2;
struct alignas(16) CalcLds { CalcNode nd[kCalcNodes]; RobotModelDev m; };

// State regularisation residual (+ its Jlog6 block) and the Euler step (+ its Jintegrate blocks, written to the workspace)
// of node tw: the scalar chain of the derivative pass.  Returns the node's state + control cost.
// PART: 0 all of it; 1 the state residual with its Jlog6 block and the two costs; 2 the Euler step with its Jintegrate blocks (the
// two halves of ik_state_kernel: independent chains of about the same length, on different waves there)
template <int PART>
__device__ __forceinline__ double node_state_terms(const IkBatchArgs &a, long b, int tw, double *ws, const IkLayout &L, const double *x,
                                                   const double *u, const double *state_w0, const double *ctrl_w0, const double *x_reg0,
                                                   const NodeTasks &tkw, double *rs, double *Jl) {
    const bool terminal_w = tw == a.T;
    const double dtw = terminal_w ? 0.0 : a.dt[b * a.T + tw];
    const double wst = tkw.state_w(), wu = tkw.ctrl_w();
    const double *state_w = state_w0 + a.sn_state_w * tw, *ctrl_w = ctrl_w0 + a.sn_ctrl_w * tw;
    double cost = 0.0;
    if (PART != 2) {
        if (wst != 0.0) {
            state_diff<true>(x_reg0 + a.sn_x_reg * tw, x, rs, Jl);
            double acc = 0.0;
            UNROLL_RBD for (int i = 0; i < kNDX; ++i) acc += state_w[i] * rs[i] * rs[i];
            cost += wst * 0.5 * acc;
        } else {
            UNROLL_RBD for (int i = 0; i < kNDX; ++i) rs[i] = 0.0;
            UNROLL_RBD for (int i = 0; i < 36; ++i) Jl[i] = (i % 7 == 0) ? 1.0 : 0.0;
        }
        if (!terminal_w) {
            double acc = 0.0;
            UNROLL_RBD for (int i = 0; i < kNV; ++i) acc += ctrl_w[i] * u[i] * u[i];
            cost += wu * 0.5 * acc;
        }
    }
    if (PART != 1 && !terminal_w) {
        double xn[kNX], A6[36], B6[36];
        euler_step<true>(x, u, dtw, xn, A6, B6);
        UNROLL_RBD for (int i = 0; i < kNX; ++i) ws[L.xnext + (long)tw * kNX + i] = xn[i];
        UNROLL_RBD for (int i = 0; i < 36; ++i) { ws[L.A6 + (long)tw * 36 + i] = A6[i]; ws[L.B6 + (long)tw * 36 + i] = B6[i]; }
    }
    return cost;
}

// The scalar chain of node tw of problem b (ik_state_kernel's lane; a lane of the fused kernel's first producer step)
template <int PART>
__device__ __forceinline__ void state_node(const IkBatchArgs &a, long b, int tw, double *ws, const IkLayout &L) {
    const int nn = a.T + 1;
    double x[kNX], u[kNV], rs[kNDX], Jl[36];
    UNROLL_RBD for (int i = 0; i < kNX; ++i) x[i] = ws[L.xs + (long)tw * kNX + i];
    UNROLL_RBD for (int i = 0; i < kNV; ++i) u[i] = tw == a.T ? 0.0 : ws[L.us + (long)tw * kNV + i];
    NodeTasks tkw{a.tasks + (b * nn + tw) * kNodeTaskDoubles};
    const double cost = node_state_terms<PART>(a, b, tw, ws, L, x, u, batch_ptr(a.state_w, a.s_state_w, b), batch_ptr(a.ctrl_w, a.s_ctrl_w, b),
                                               a.x_reg + b * a.s_x_reg, tkw, rs, Jl);
    if (PART != 2) {
        UNROLL_RBD for (int i = 0; i < kNDX; ++i) ws[L.nrs + (long)tw * kNDX + i] = rs[i];
        UNROLL_RBD for (int i = 0; i < 6; ++i) UNROLL_RBD for (int k = 0; k < 6; ++k) ws[L.njl + (long)tw * 36 + 6 * i + k] = Jl[6 * k + i];
        ws[L.ncs + tw] = cost;
    }
}

// ... for all nodes of all problems, ONE LANE PER NODE, launched before ik_calcdiff_kernel.  Inside that kernel (where it used
// to run, on lanes 0 and 32 of wave 1) this chain kept two lanes of a wave busy, and in the bulk iterations -- where
// calcdiff is bound by instruction issue -- those two lanes cost a quarter of the workgroup's issue slots; here 64 nodes
// share every instruction.  One code path whatever the number of active problems, so a problem's results do not depend
// on its batch (the few microseconds of an extra launch per tail iteration are the price).
__global__ __launch_bounds__(64) void ik_state_kernel(const IkBatchArgs a) {
    const int nn = a.T + 1;
    // even workgroups: the state residual, its Jlog6 block, the costs; odd workgroups: the Euler step and its Jintegrate blocks -- two
    // independent chains of a node, each about half of the lane's work.  The kernel is latency-bound in every regime (one wave per
    // SIMD at 434 registers, never more waves than SIMDs): half the chain is half the time.
    const long idx = (long)(blockIdx.x >> 1) * 64 + threadIdx.x;
    const bool second = (blockIdx.x & 1) != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.count) { a.count[(a.iter + 1) & 1] = 0; a.wcount[(a.iter + 1) & 1] = 0; a.near[(a.iter + 1) & 1] = 0; }     // the lists this iteration's forward pass will fill
    const long b = slot_problem(a, idx / nn);
    const int tw = (int)(idx % nn);
    if (b < 0) return;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    if (ws[L.scal + S_DONE] != 0.0 || ws[L.scal + S_RECALC] == 0.0) return;
    if (second) state_node<2>(a, b, tw, ws, L);
    else state_node<1>(a, b, tw, ws, L);
}

// ---- the pieces of the derivative pass of a node, shared by ik_calcdiff_kernel (two waves for two nodes) and the fused
// kernel's producer waves (one wave for two nodes, the pieces one after the other): same code, same arithmetic, same bits.
// (1) a lane's part of the robot walk (lanes hl < 18 of a node's half-wave): parts published to the node's LDS block
__device__ __forceinline__ void calc_walk(const RobotModelDev &m, CalcNode &qw, const NodeTasks &tkw, int hl, double (&Rb)[9], double (&pb)[3],
                                          double (&Vb)[6], PartWalk &pw) {
    int fid[kFrameSlots];
    UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) fid[f] = tkw.frame_w(f) != 0.0 ? tkw.frame_id(f) : -1;
    quad_part_walk(m, qw.x, fid, hl, Rb, pb, Vb, pw);
    const bool leg_pub = hl >= 6 && (hl - 6) % kLegJoints == 0;
    if (hl == 0 || leg_pub) {   // one publisher per part
        double *pp = qw.parts[hl == 0 ? kLegs : (hl - 6) / kLegJoints];
        pp[0] = pw.part.m;
        UNROLL_RBD for (int c = 0; c < 3; ++c) pp[1 + c] = pw.part.h1[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) { pp[4 + c] = pw.part.I[c]; pp[10 + c] = pw.hpart[c]; }
        UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
            if (pw.fhit[f]) { UNROLL_RBD for (int c = 0; c < 3; ++c) qw.fx[f][c] = pw.fx[f][c]; }
    }
}
// (2) what ik_state_kernel / state_node left for node tw: state residual, its Jlog6 block (transposed), the state + control cost
__device__ __forceinline__ void calc_fetch_state(CalcNode &qw, const double *ws, const IkLayout &L, int tw, int hl) {
    for (int i = hl; i < kNDX; i += 32) {
        qw.rs[i] = ws[L.nrs + (long)tw * kNDX + i];
        (&qw.JlT[0][0])[i] = ws[L.njl + (long)tw * 36 + i];      // stored transposed already
    }
    if (hl == 0) qw.cost_sc = ws[L.ncs + tw];
}
// (3) robot totals from the five parts, then this lane's column of the residual Jacobian (and the residuals on lane 0)
__device__ __forceinline__ void calc_columns(const RobotModelDev &m, CalcNode &qw, const NodeTasks &tkw, int hl, PartWalk &pw) {
    Comp call; double hO[6];
    {
        double2_t pa[20], pb2[20];
        lds_read_b128x20(lds_offset(qw.parts), pa);
        lds_read_b128x20(lds_offset(qw.parts) + 320, pb2);
        double tot[kPartDoubles];
        UNROLL_RBD for (int k = 0; k < kPartDoubles; ++k) {
            double acc = 0.0;
            UNROLL_RBD for (int q = 0; q < kParts; ++q) {
                const int e = q * kPartDoubles + k;   // element index in the 80-double block
                const double2_t v2 = e < 40 ? pa[e >> 1] : pb2[(e - 40) >> 1];
                acc += (e & 1) ? v2.y : v2.x;
            }
            tot[k] = acc;
        }
        call.m = tot[0];
        UNROLL_RBD for (int c = 0; c < 3; ++c) call.h1[c] = tot[1 + c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) { call.I[c] = tot[4 + c]; hO[c] = tot[10 + c]; }
    }
    const double M = call.m, iM = 1.0 / M;
    double com[3], t3[3], hg[6];
    UNROLL_RBD for (int c = 0; c < 3; ++c) com[c] = call.h1[c] * iM;
    cross3(com, hO, t3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) { hg[c] = hO[c]; hg[3 + c] = hO[3 + c] - t3[c]; }
    if (hl < 6) { pw.cs = call; UNROLL_RBD for (int c = 0; c < 6; ++c) pw.hs[c] = hO[c]; }   // base columns move the whole robot
    Column c;
    quad_col_finish(pw, M, com, hO, c);
    double *cq = qw.Jt[hl], *cv = qw.Jt[kNV + hl];
    UNROLL_RBD for (int k = 0; k < 6; ++k) { cq[k] = c.dh[k]; cv[k] = c.ag[k]; }
    UNROLL_RBD for (int k = 0; k < 3; ++k) cq[6 + k] = c.jc[k];
    UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) {
        double j3[3] = {0, 0, 0};
        if (tkw.frame_w(f) != 0.0 && quad_supports(m, tkw.frame_id(f), hl)) {
            const double fxf[3] = {qw.fx[f][0], qw.fx[f][1], qw.fx[f][2]};
            cross3(c.S + 3, fxf, j3);
            UNROLL_RBD for (int k = 0; k < 3; ++k) j3[k] += c.S[k];
        }
        UNROLL_RBD for (int k = 0; k < 3; ++k) cq[9 + 3 * f + k] = j3[k];
    }
    cq[kRes] = 0.0;
    UNROLL_RBD for (int k = 6; k < kResLd; ++k) cv[k] = 0.0;
    if (hl == 0) {   // residuals and their cost
        double cost = 0.0, acc = 0.0;
        UNROLL_RBD for (int k = 0; k < 6; ++k) { const double rr = hg[k] - tkw.mom_ref()[k]; qw.res[k] = rr; acc += rr * rr; }
        cost += tkw.mom_w() * 0.5 * acc;
        acc = 0.0;
        UNROLL_RBD for (int k = 0; k < 3; ++k) { const double rr = com[k] - tkw.com_ref()[k]; qw.res[6 + k] = rr; acc += rr * rr; }
        cost += tkw.com_w() * 0.5 * acc;
        UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) {
            const double w = tkw.frame_w(f);
            acc = 0.0;
            UNROLL_RBD for (int k = 0; k < 3; ++k) { const double rr = w != 0.0 ? qw.fx[f][k] - tkw.frame_ref(f)[k] : 0.0; qw.res[9 + 3 * f + k] = rr; acc += rr * rr; }
            cost += w * 0.5 * acc;
        }
        qw.res[kRes] = 0.0;
        qw.cost_kin = cost;
    }
}
// (4) Gauss-Newton L_x / L_xx / L_u / L_uu / cost of node t from its LDS block, by ONE wave (all 64 lanes)
__device__ __forceinline__ void calc_assemble(const IkBatchArgs &a, long b, int t, CalcNode &q, double *ws, const IkLayout &L, int lane,
                                              const double *state_w0, const double *ctrl_w0, double *cost_slot) {
    NodeTasks tk{q.tk};
    const bool terminal = t == a.T;
    const double dt = q.tk[kNodeTaskDoubles];
    const double wm = tk.mom_w(), wc = tk.com_w(), wst = tk.state_w(), wu = tk.ctrl_w();
    // The weight vectors and the workspace through GLOBAL-address-space pointers (in the non-inlined instance the arguments come out
    // of LDS and the pointers are generic: flat loads and stores, which count against the LDS counter too and were each waited for
    // on their own), and every weight a lane needs requested here, in one batch, before anything waits.
    typedef const double __attribute__((address_space(1))) *gcd_t;
    typedef double __attribute__((address_space(1))) *gd_t;
    const gcd_t state_w = (gcd_t)(state_w0 + a.sn_state_w * t), ctrl_w = (gcd_t)(ctrl_w0 + a.sn_ctrl_w * t);
    const gd_t wsg = (gd_t)ws;
    const int li_ = lane & 15;
    double sw_k[6];
    UNROLL_RBD for (int k = 0; k < 6; ++k) sw_k[k] = state_w[k];
    const double sw_j = state_w[lane < kNDX ? lane : 0], sw_li = state_w[li_], sw_c = state_w[16 + (li_ < 2 ? li_ : 0)];
    const double sw_t = state_w[16 + (lane < 20 ? lane : 0)], cw_i = ctrl_w[lane >= 40 && lane < 40 + kNV ? lane - 40 : 0];
    const double sc = terminal ? 1.0 : dt;
    if (!terminal && lane >= 40 && lane < 40 + kNV) {     // lanes the assembly leaves idle
        const int i = lane - 40;
        wsg[L.Lu + (long)t * kNV + i] = sc * wu * cw_i * q.u[i];
        wsg[L.Luu + (long)t * kNV + i] = sc * wu * cw_i;
    }
    // node costs are summed by the backward pass: parked in the gap slot of this node (multi-kernel path) / in LDS (fused kernel)
    if (lane == 63) *cost_slot = terminal ? q.cost_kin + q.cost_sc : dt * (q.cost_kin + q.cost_sc);
    // Gauss-Newton L_x (lane j = entry j): the weighted own column of the residual Jacobian against the residuals.
    const int j = lane < kNDX ? lane : 0;
    if (lane < kNDX) {
        double jw[kRes];
        {
            double2_t own[11];
            lds_read_b128x11(lds_offset(q.Jt[j]), own);
            UNROLL_RBD for (int k = 0; k < kRes; ++k) jw[k] = (k & 1) ? own[k >> 1].y : own[k >> 1].x;
        }
        {   // the momentum Jacobian M = d h_g / d (q, v), row-major, for the Riccati pass (IkLayout: kHnDoubles): lane j = column j
            const gd_t Hn = wsg + L.Hn + (long)t * kHnDoubles;
            UNROLL_RBD for (int k = 0; k < 6; ++k) Hn[k * kNDX + j] = jw[k];
        }
        UNROLL_RBD for (int k = 0; k < 6; ++k) jw[k] *= wm;
        UNROLL_RBD for (int k = 0; k < 3; ++k) jw[6 + k] *= wc;
        UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
            UNROLL_RBD for (int k = 0; k < 3; ++k) jw[9 + 3 * f + k] *= tk.frame_w(f);
        double2_t rr[11];
        lds_read_b128x11(lds_offset(q.res), rr);
        double g = 0.0;
        UNROLL_RBD for (int k = 0; k < kRes; ++k) g += jw[k] * ((k & 1) ? rr[k >> 1].y : rr[k >> 1].x);
        if (j < 6) { UNROLL_RBD for (int k = 0; k < 6; ++k) g += wst * sw_k[k] * q.JlT[j][k] * q.rs[k]; }
        else g += wst * sw_j * q.rs[j];
        wsg[L.Lx + (long)t * kNDX + j] = sc * g;
    }
    // Gauss-Newton L_xx = sc J^T W J (+ the state regularisation).  Only its q-block WITHOUT the momentum rows is formed here:
    //     L_qq' = J_c^T wc J_c + sum_f J_f^T w_f J_f + (state regularisation on q: Jlog6 block on the free-flyer, weights on the joints)
    // on the matrix pipe, three tiles of v_mfma_f64_16x16x4 over the residual rows 6..20 (k-steps 1..5; the momentum rows 0..5 get
    // weight zero), the rows of Jt in LDS serving as both operands (A weighted).  Lane l holds A[l & 15][k = l >> 4], B[k][l & 15] and
    // D[(l >> 4) + 4 v][l & 15].  The momentum term wm M^T M -- the only part of L_xx that reaches the velocity columns -- is added
    // by the Riccati pass itself from M (stored above), in the same tile layout its Schur update runs in; the tiles go to the
    // workspace AS TILES (IkLayout: kLqqDoubles), each store instruction one contiguous 512 / 64 / 32 bytes.
    {
        const int li = lane & 15, lk = lane >> 4;
        double av[18], bv[18];      // [6 I + ks]: Jt[16 I + li][4 ks + lk]
        lds_read_calc_operand(lds_offset(&q.Jt[0][0]) + (unsigned)(li * kResLd + lk) * 8u, bv);
        UNROLL_RBD for (int ks = 1; ks < 6; ++ks) {
            const int k = 4 * ks + lk;                       // residual row this lane feeds in k-step ks
            const int f = k >= 9 ? (k - 9) / 3 : 0;
            const double fw = f == 0 ? tk.frame_w(0) : f == 1 ? tk.frame_w(1) : f == 2 ? tk.frame_w(2) : tk.frame_w(3);
            const double w = k < 6 ? 0.0 : k < 9 ? wc : k < kRes ? fw : 0.0;
            UNROLL_RBD for (int I = 0; I < 2; ++I) {
                if (ks == 5) bv[6 * I + ks] = k < kRes ? bv[6 * I + ks] : 0.0;     // columns 21..23 are padding / the next row
                av[6 * I + ks] = w * bv[6 * I + ks];
            }
        }
        const gd_t Lqq = wsg + L.Lqq + (long)t * kLqqDoubles;
        mfma_acc_t a00 = mfma_acc_t{0.0, 0.0, 0.0, 0.0}, a01 = a00, a11 = a00;
        UNROLL_RBD for (int ks = 1; ks < 6; ++ks) {
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[6 + ks], a01, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[6 + ks], bv[6 + ks], a11, 0, 0, 0);
        }
        UNROLL_RBD for (int ks = 0; ks < 2; ++ks) {  // the Jlog6 block: rows / columns 0..5 of tile (0, 0)
            const int k = 4 * ks + lk;
            const bool ok = li < 6 && k < 6;
            double swk = sw_k[0];       // state_w[k] of this lane's row (k = lk or 4 + lk), out of the six wave-uniform values
            UNROLL_RBD for (int c = 1; c < 6; ++c) swk = (ok && k == c) ? sw_k[c] : swk;
            const double jl = q.JlT[ok ? li : 0][ok ? k : 0];
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? wst * swk * jl : 0.0, ok ? jl : 0.0, a00, 0, 0, 0);
        }
        const double sw0 = li >= 6 ? wst * sw_li : 0.0;            // the joints' part of the diagonal of tile (0, 0)
        const double sw1 = wst * sw_c;                              // ... of the 2 x 2 corner of tile (1, 1) (q 16, 17)
        UNROLL_RBD for (int v = 0; v < 4; ++v) Lqq[v * 64 + lane] = sc * (a00[v] + (lk + 4 * v == li ? sw0 : 0.0));
        if (li < 2) { UNROLL_RBD for (int v = 0; v < 4; ++v) Lqq[256 + (v * 4 + lk) * 2 + li] = sc * a01[v]; }
        if (li < 2 && lk < 2) Lqq[288 + lk * 2 + li] = sc * (a11[0] + (lk == li ? sw1 : 0.0));
        // the momentum weight and the velocity part of the state regularisation's diagonal, as the tiles (1,1) / (2,2) index it
        const gd_t Hn = wsg + L.Hn + (long)t * kHnDoubles;
        if (lane < 16) Hn[kHnD11 + lane] = lane >= 2 ? sc * wst * sw_t : 0.0;
        else if (lane < 20) Hn[kHnD22 + lane - 16] = sc * wst * sw_t;
        else if (lane == 20) Hn[kHnW] = sc * wm;
    }
}

// Workgroup = two waves for TWO nodes of a problem.  Wave 0: lanes 0..17 (node A) and 32..49 (node B) each walk their
// own part of the robot once (base lanes the base body, joint lanes their leg), the five part sums of a node meet in
// LDS, every lane finishes its column.  Wave 1, meanwhile, fetches what ik_state_kernel left for the two nodes (a different
// instruction stream, so it would serialise inside wave 0).  Then wave w assembles the Gauss-Newton L_x / L_xx of node w,
// lane j = column j.  Two nodes share every instruction of the walk: the walk keeps 36 of 64 lanes busy instead of 18.
__global__ __launch_bounds__(128, 2) void ik_calcdiff_kernel(const IkBatchArgs a) {
    __shared__ CalcLds s;
    const int nn = a.T + 1, groups = (nn + kCalcNodes - 1) / kCalcNodes;
    const long b = slot_problem(a, blockIdx.x / groups);
    if (b < 0) return;
    const int t0 = (blockIdx.x % groups) * kCalcNodes, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    if (ws[L.scal + S_DONE] != 0.0 || ws[L.scal + S_RECALC] == 0.0) return;
#ifdef BWD_PROFILE
    long long pc[6] = {0, 0, 0, 0, 0, 0}, pt0 = __builtin_readcyclecounter();
#endif
    {
        stage_model<128>(a.model, &s.m);
    }
    const RobotModelDev &m = s.m;
    const double *state_w0 = batch_ptr(a.state_w, a.s_state_w, b), *ctrl_w0 = batch_ptr(a.ctrl_w, a.s_ctrl_w, b);
    {   // states and controls of both nodes: wave w stages node w
        const int t = t0 + wave;
        if (t < nn) { const StagedNode v = calc_stage_load(a, b, t, ws, L, lane); calc_stage_store(a, t, s.nd[wave], lane, v); }
    }
    __syncthreads();
    PSTAMP(0)
    // ---- walk phase: node index = half of the wave
    const int hs = lane >> 5, hl = lane & 31, tw = t0 + hs;
    const bool wvalid = tw < nn;
    CalcNode &qw = s.nd[hs];
    NodeTasks tkw{qw.tk};
    PartWalk pw;
    double Rb[9], pb[3], Vb[6];
    if (wave == 0 && wvalid && hl < kNV) calc_walk(m, qw, tkw, hl, Rb, pb, Vb, pw);
    else if (wave == 1 && wvalid) calc_fetch_state(qw, ws, L, tw, hl);
    __syncthreads();
    PSTAMP(1)
    if (wave == 0 && wvalid && hl < kNV) calc_columns(m, qw, tkw, hl, pw);
    __syncthreads();
    PSTAMP(2)
    // ---- assembly phase: wave w owns node t0 + w
    const int t = t0 + wave;
    if (t >= nn) return;
    calc_assemble(a, b, t, s.nd[wave], ws, L, lane, state_w0, ctrl_w0, ws + L.fs + (long)t * kNDX);
#ifdef BWD_PROFILE
    PSTAMP(3)
    if (threadIdx.x == 0 && t == 0) { for (int k = 0; k < 4; ++k) ws[L.Qu + k] = (double)pc[k]; }   // the Qu slot is unused by the solver
#endif
}

// The same derivative pass by ONE wave per node pair, the pieces one after the other (as the fused kernel's producer waves run
// them): for launches with more workgroups than the chip has wave slots.  There the two-wave kernel is bound by the slots it
// holds, and its second wave holds one through the walk only to wait at the barrier; a single wave spends 37 % fewer
// wave-cycles per pair (the pair's latency, which is what counts when few problems are left, is 25 % longer).
// Two such waves share a workgroup (and its copy of the robot model), each with a node pair of its own: 36.7 KB of LDS per
// workgroup, four workgroups = eight waves per CU, which is what the registers allow.  The assembly of a node is a NON-INLINED
// function (as the fused kernel's roles, further down): in one body with the walk, hipcc spilled 52 registers at the kernel's
// 256 (122 under -ffp-contract=on); on its own the assembly has its own allocation, and at its call the walk's state is dead.
struct alignas(16) Calc1Lds { CalcNode nd[2][kCalcNodes]; RobotModelDev m; IkBatchArgs args; };
__shared__ Calc1Lds g_calc1;
__device__ __forceinline__ IkBatchArgs uniform_args(const IkBatchArgs &g);      // every field through v_readfirstlane (defined with the fused kernel)
__device__ __forceinline__ int uni(int v);
__device__ __forceinline__ long uni(long v);
__device__ __noinline__ void calc1_assemble(long b_, int t_, int h_) {
    const IkBatchArgs a = uniform_args(g_calc1.args);
    const long b = uni(b_);
    const int t = uni(t_), h = uni(h_), lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    calc_assemble(a, b, t, g_calc1.nd[wave][h], ws, L, lane, batch_ptr(a.state_w, a.s_state_w, b), batch_ptr(a.ctrl_w, a.s_ctrl_w, b),
                  ws + L.fs + (long)t * kNDX);
}
#ifndef CALC1_WPE
#define CALC1_WPE 2
#endif
__global__ __launch_bounds__(128, CALC1_WPE) void ik_calcdiff1_kernel(const IkBatchArgs a) {
    Calc1Lds &s = g_calc1;
    const int nn = a.T + 1, groups = (nn + kCalcNodes - 1) / kCalcNodes;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        stage_model<128>(a.model, &s.m);
        if (threadIdx.x == 0) s.args = a;
    }
    __syncthreads();        // (the only workgroup barrier: from here on the two waves have nothing to do with each other)
    const long unit = (long)blockIdx.x * 2 + wave;
    const long b = slot_problem(a, unit / groups);
    if (b < 0) return;
    const int t0 = (int)(unit % groups) * kCalcNodes;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    const RobotModelDev &m = s.m;
    CalcNode (&nd)[kCalcNodes] = s.nd[wave];
    {   // the problem's two flags and everything the pair needs from the workspace in ONE round trip (the flags first and the states
        // behind their test were two)
        const double f_done = ws[L.scal + S_DONE], f_recalc = ws[L.scal + S_RECALC];
        const bool two = t0 + 1 < nn;
        const StagedNode v0 = calc_stage_load(a, b, t0, ws, L, lane), v1 = calc_stage_load(a, b, two ? t0 + 1 : t0, ws, L, lane);
        if (f_done != 0.0 || f_recalc == 0.0) return;
        calc_stage_store(a, t0, nd[0], lane, v0);
        if (two) calc_stage_store(a, t0 + 1, nd[1], lane, v1);
    }
    wave_sync();
    const int hs = lane >> 5, hl = lane & 31, tw = t0 + hs;
    const bool wvalid = tw < nn;
    CalcNode &qw = nd[hs];
    {
        NodeTasks tkw{qw.tk};
        PartWalk pw;
        double Rb[9], pb[3], Vb[6];
        if (wvalid && hl < kNV) calc_walk(m, qw, tkw, hl, Rb, pb, Vb, pw);
        if (wvalid) calc_fetch_state(qw, ws, L, tw, hl);
        wave_sync();
        if (wvalid && hl < kNV) calc_columns(m, qw, tkw, hl, pw);
    }
    wave_sync();
    calc1_assemble(b, t0, 0);
    if (t0 + 1 < nn) calc1_assemble(b, t0 + 1, 1);
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
constexpr int kQuLane = kNDX + kNV;   // backward pass: lanes 0..35 x-rows, 36..53 rows of Q_uu, 54 the Q_u row
// The Schur update V_xx = Q_xx - Q_xu K = Q_xx - Y^T Y (Y = L^-1 Q_ux) runs on the matrix pipe as tiles of v_mfma_f64_16x16x4
// (36 -> 48, k: 18 -> 20).  Operand image in LDS: Y^T as [48][LDK] (zero outside 36 x 18, so the padding contributes nothing),
// Q_xx / V_xx in the rows of N ([48][LD]: the padded rows are read and written but never used).  Lane l holds A[l & 15][k = l >> 4],
// B[k = l >> 4][l & 15] and D[(l >> 4) + 4 v][l & 15], v < 4; every element a lane touches sits at a compile-time offset
// from one per-lane base address, so the reads are batches of ds_read_b64 with immediate offsets (lds_batch.h explains
// why they are asm: hipcc puts a full wait after each LDS read it schedules itself).
constexpr int LDK = 21, kPadRows = 48;
#define BWD_STR2(x) #x
#define BWD_STR(x) BWD_STR2(x)
#define BWD_OFFK(I, ks) ((16 * I * 21 + 4 * ks) * 8)
#define BWD_OFFN(I, J, v) (((16 * I + 4 * v) * 37 + 16 * J) * 8)
#define BWD_RDK(n, I, ks) "ds_read_b64 %" #n ", %15 offset:" BWD_STR(BWD_OFFK(I, ks)) "\n"
#define BWD_RDN(n, I, J, v) "ds_read_b64 %" #n ", %12 offset:" BWD_STR(BWD_OFFN(I, J, v)) "\n"
// o[5 I + ks] = image[16 I + (l & 15)][4 ks + (l >> 4)]
__device__ __forceinline__ void lds_read_mfma_operand(unsigned base, double (&o)[15]) {
    asm volatile(BWD_RDK(0, 0, 0) BWD_RDK(1, 0, 1) BWD_RDK(2, 0, 2) BWD_RDK(3, 0, 3) BWD_RDK(4, 0, 4)
                 BWD_RDK(5, 1, 0) BWD_RDK(6, 1, 1) BWD_RDK(7, 1, 2) BWD_RDK(8, 1, 3) BWD_RDK(9, 1, 4)
                 BWD_RDK(10, 2, 0) BWD_RDK(11, 2, 1) BWD_RDK(12, 2, 2) BWD_RDK(13, 2, 3) BWD_RDK(14, 2, 4)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14])
                 : "v"(base) : "memory");
}
// the accumulator tiles on and above the block diagonal, the order the Schur update numbers them in:
// tile 0..5 = (0,0) (0,1) (0,2) (1,1) (1,2) (2,2);  o[4 tile + v] = N[16 I + (l >> 4) + 4 v][16 J + (l & 15)]
__device__ __forceinline__ void lds_read_mfma_acc_upper(unsigned base, double (&o)[24]) {
    asm volatile(BWD_RDN(0, 0, 0, 0) BWD_RDN(1, 0, 0, 1) BWD_RDN(2, 0, 0, 2) BWD_RDN(3, 0, 0, 3)
                 BWD_RDN(4, 0, 1, 0) BWD_RDN(5, 0, 1, 1) BWD_RDN(6, 0, 1, 2) BWD_RDN(7, 0, 1, 3)
                 BWD_RDN(8, 0, 2, 0) BWD_RDN(9, 0, 2, 1) BWD_RDN(10, 0, 2, 2) BWD_RDN(11, 0, 2, 3)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
                 : "v"(base) : "memory");
    asm volatile(BWD_RDN(0, 1, 1, 0) BWD_RDN(1, 1, 1, 1) BWD_RDN(2, 1, 1, 2) BWD_RDN(3, 1, 1, 3)
                 BWD_RDN(4, 1, 2, 0) BWD_RDN(5, 1, 2, 1) BWD_RDN(6, 1, 2, 2) BWD_RDN(7, 1, 2, 3)
                 BWD_RDN(8, 2, 2, 0) BWD_RDN(9, 2, 2, 1) BWD_RDN(10, 2, 2, 2) BWD_RDN(11, 2, 2, 3)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19]),
                   "=&v"(o[20]), "=&v"(o[21]), "=&v"(o[22]), "=&v"(o[23])
                 : "v"(base) : "memory");
}

#ifndef BWD_WPE
#define BWD_WPE 1      // waves per SIMD the one-wave Riccati kernel is compiled for (2 = the 256-register build)
#endif
constexpr bool kBwdLean = BWD_WPE == 2;
// lean build: nothing moves across a phase boundary of the node (hipcc otherwise stretches the phases over each other -- the staged
// row of G stays in registers through the elimination, the tiles' operands are fetched during the back substitution, ...)
#define LEAN_FENCE if (NWB == 1 && kBwdLean) __builtin_amdgcn_sched_barrier(0);     // the one-wave kernel's low-register variant (same arithmetic, fewer values alive at once)
// a wave-uniform double moved to scalar registers
__device__ __forceinline__ double uni_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// value of v in lane `src` (compile-time constant), wave-uniform: two v_readlane_b32
__device__ __forceinline__ double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// 1 / sqrt(a) to ~1 ulp: v_rsq_f64 + two Newton steps (a > 0; NaN for a < 0, which the caller has flagged by then)
__device__ __forceinline__ double rsqrt64(double a) {
    double r = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    r = fma(fma(-h * r, r, 0.5), r, r);
    return fma(fma(-h * r, r, 0.5), r, r);
}
// 1 / b to ~1 ulp: v_rcp_f64 + two Newton steps
__device__ __forceinline__ double rcp64(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    return fma(fma(-b, r, 1.0), r, r);
}

// What a lane needs of node t's compact L_xx (IkLayout: Lqq tiles, Hn) for its elements of the six upper MFMA tiles: 18 loads,
// each wave instruction over contiguous bytes (the full row it replaces: 36 loads with a 288-byte lane stride).
struct LxxLane { double lq[9], mb[6], w, d11, d22; };
// element of a global array addressed as a WAVE-UNIFORM base (scalar registers) + a 32-bit per-lane byte offset: `global_load v,
// v_off, s[base:base+1] offset:imm`.  Written as `ws + L.x + t * n + lane` the address is a 64-bit per-lane sum, and the Riccati
// node keeps some twenty such pointer pairs alive (or, in the 256-register build, reloads them from scratch at every use).
__device__ __forceinline__ const double *at_u(const double *ubase, unsigned byte_off) {
    return reinterpret_cast<const double *>(reinterpret_cast<const char *>(ubase) + byte_off);
}
__device__ __forceinline__ double *at_u(double *ubase, unsigned byte_off) {
    return reinterpret_cast<double *>(reinterpret_cast<char *>(ubase) + byte_off);
}
// the per-lane byte offsets of lxx_lane_load (node-independent: computed once per pass)
struct LxxOffsets { unsigned t00, t01, t11, m0, m1, m02, m12, d11, d22; };
__device__ __forceinline__ LxxOffsets lxx_offsets(int lane) {
    const unsigned li = lane & 15, lk = lane >> 4, c = li < 2 ? li : 0;
    LxxOffsets o;
    o.t00 = 8u * (unsigned)lane;                                  // Lqq[v * 64 + lane]
    o.t01 = 8u * (256u + lk * 2u + c);                            // Lqq[256 + (v * 4 + lk) * 2 + c]
    o.t11 = 8u * (288u + (lk & 1u) * 2u + c);
    // M[k = 4 ks + lk][j = 16 J + li]: every lane loads an existing element (what is not its own becomes zero after the load)
    o.m0 = 8u * (lk * (unsigned)kNDX + li);                       // ks = 0, J = 0, 1 (+ 16 J in the immediate)
    o.m02 = li < 4 ? 8u * (lk * (unsigned)kNDX + 32u + li) : 0u;  // ks = 0, J = 2
    o.m1 = lk < 2 ? 8u * ((4u + lk) * (unsigned)kNDX + li) : 0u;  // ks = 1, J = 0, 1
    o.m12 = lk < 2 && li < 4 ? 8u * ((4u + lk) * (unsigned)kNDX + 32u + li) : 0u;
    o.d11 = 8u * ((unsigned)kHnD11 + li);
    o.d22 = 8u * ((unsigned)kHnD22 + (li & 3u));
    return o;
}
// wsu: the problem's workspace (wave-uniform)
__device__ __forceinline__ void lxx_lane_load(const double *wsu, const IkLayout &L, int t, int lane, const LxxOffsets &f, LxxLane &o) {
    const int li = lane & 15, lk = lane >> 4;
    const double *Lqq = wsu + L.Lqq + (long)t * kLqqDoubles, *Hn = wsu + L.Hn + (long)t * kHnDoubles;
    UNROLL_RBD for (int v = 0; v < 4; ++v) o.lq[v] = at_u(Lqq, f.t00)[v * 64];
    UNROLL_RBD for (int v = 0; v < 4; ++v) { const double x = at_u(Lqq, f.t01)[v * 8]; o.lq[4 + v] = li < 2 ? x : 0.0; }
    { const double x = *at_u(Lqq, f.t11); o.lq[8] = li < 2 && lk < 2 ? x : 0.0; }
    {
        const double x00 = at_u(Hn, f.m0)[0], x01 = at_u(Hn, f.m0)[16], x02 = *at_u(Hn, f.m02);
        const double x10 = at_u(Hn, f.m1)[0], x11 = at_u(Hn, f.m1)[16], x12 = *at_u(Hn, f.m12);
        o.mb[0] = x00; o.mb[1] = x01; o.mb[2] = li < 4 ? x02 : 0.0;
        o.mb[3] = lk < 2 ? x10 : 0.0; o.mb[4] = lk < 2 ? x11 : 0.0; o.mb[5] = lk < 2 && li < 4 ? x12 : 0.0;
    }
    o.w = Hn[kHnW];
    o.d11 = *at_u(Hn, f.d11);
    { const double x = *at_u(Hn, f.d22); o.d22 = li < 4 ? x : 0.0; }
}
// acc (the six tiles on and above the block diagonal, numbered as in the Schur update) += L_xx of the node
__device__ __forceinline__ void lxx_add_tiles(mfma_acc_t (&acc)[6], const LxxLane &x, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    UNROLL_RBD for (int v = 0; v < 4; ++v) {
        acc[0][v] += x.lq[v];
        acc[1][v] += x.lq[4 + v];
        acc[3][v] += lk + 4 * v == li ? x.d11 : 0.0;          // velocity diagonal inside tile (1, 1): columns 18..31
    }
    acc[3][0] += x.lq[8];
    acc[5][0] += lk == li ? x.d22 : 0.0;                     // ... inside tile (2, 2): columns 32..35
    UNROLL_RBD for (int ks = 0; ks < 2; ++ks)                 // wm M^T M: six momentum rows = two k-steps (rows 6, 7 are zero)
        UNROLL_RBD for (int I = 0; I < 3; ++I)
            UNROLL_RBD for (int J = I; J < 3; ++J) {
                const int tl = (I == 0 ? 0 : I == 1 ? 2 : 3) + J;
                acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.w * x.mb[3 * ks + I], x.mb[3 * ks + J], acc[tl], 0, 0, 0);
            }
}
// the six tiles -> the staged rows N (mirror image of the off-diagonal tiles too), xreg joining the diagonal
__device__ __forceinline__ void tiles_to_rows(double *N, const mfma_acc_t (&acc)[6], double xreg, int lane);

__device__ __forceinline__ void tiles_to_rows(double *N, const mfma_acc_t (&acc)[6], double xreg, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    UNROLL_RBD for (int I = 0; I < 3; ++I)
        UNROLL_RBD for (int J = I; J < 3; ++J)
            if (J < 2 || li < kNDX - 32) {      // columns 36..47 do not exist (rows 36..47 do, as padding)
                const int tl = (I == 0 ? 0 : I == 1 ? 2 : 3) + J;
                UNROLL_RBD for (int v = 0; v < 4; ++v) {
                    const int i = 16 * I + lk + 4 * v, j = 16 * J + li;
                    // V = (V_xx + V_xx^T)/2 + xreg I: xreg joins the staged diagonal here (a diagonal entry averages with itself)
                    N[i * LD + j] = (J == I && i == j) ? acc[tl][v] + xreg : acc[tl][v];
                    if (J > I) N[j * LD + i] = acc[tl][v];
                }
            }
}

// NB = 2 (the two-wave kernel): what the gains wave needs of a node -- the factor, Y^T, y_u, the reciprocal pivots, Q_u -- is
// double-buffered, node t in buffer t & 1 of the pass' running count, so the recursion writes node t - 1 while the gains of
// node t are still being read.
template <int NB>
struct alignas(16) BackwardLds {
    double N[kPadRows * LD + 16];   // row-major staging: N = F_x^T V for the transposition, later Q_xx -> V_xx rows
    double Ys[NB][kPadRows * LDK];  // Y^T, Y = L^-1 Q_ux (36 x 18 in a zeroed 48 x 21 image): both operands of the Schur update
    double Lc[NB][5 * 36];          // Cholesky factor packed by rows (L[p][q], q < p, at p(p-1)/2 + q), read back by broadcast in
                                    // batches of 36
    double A6[36], B6[36];
    double Vx[kNDX], fs[kNDX];
    double yu[NB][kNV + 2], idg[NB][kNV + 2], qu[NB][kNV + 2];     // y_u = L^-1 Q_u, 1 / L[p][p], Q_u   (gains wave only)
    int tnode[NB];                  // which node the buffer holds; -1: the pass is over
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

// NWB = 2 (few problems still iterating, EXPERIMENTS.md 9): the gains K = L^-T Y, k = L^-T y_u -- the back substitutions, their
// stores and the terms of the expected improvement that need k -- are for the forward pass, not for the next node of the
// recursion (V_xx = Q_xx - Y^T Y and V_x = Q_x - Y^T y_u come from the forward substitutions alone).  A second wave takes
// them over, one node behind: the recursion hands it the factor and Y through LDS and goes on.
__device__ __forceinline__ void bwd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- control block of the fused kernel (ik_fused_kernel, below): one persistent four-wave workgroup per problem that runs
// whole DDP iterations without leaving the chip.  Its derivative pass and Riccati pass share ONE sequence of workgroup barriers
// ("ticks"): wave 0 runs the recursion (a node per tick), wave 1 the gains one node behind, waves 2 and 3 produce the node
// derivatives ahead of the recursion.  Every wave executes exactly one barrier per tick whatever it has to do in it (a node, a
// producer step, or nothing), and all of them leave the phase after the same barrier: the gains wave announces the NUMBER of the
// last tick (done_tick) and a wave leaves once its own count has reached it.  (A plain "done" flag is not enough: set between
// barriers n and n + 1, it can be seen by a wave that reads it late after barrier n and by another only after n + 1 -- they
// would leave one barrier apart and every later barrier of the kernel would pair up wrongly.  That version ran clean for a day
// and then, with a changed code layout, handed stale node costs to the line search of one problem in twenty.)
// Flags compare against `stamp` (bumped once per phase), so nothing has to be cleared.
struct FusedCtl {
    int ready[64];          // node t's derivatives are in the workspace (stamp)
    int state_ready;        // ... the scalar chains of every node (state residual, Euler step: state_node) (stamp)
    int hand_count;         // nodes (and end markers) the recursion has handed to the gains wave in this phase
    int done_tick;          // the phase ends with the tick of this number (INT_MAX while it is open): set by the gains wave
    int stamp;
    double node_cost[64];   // the node costs (the multi-kernel path parks them in the nodes' gap slots)
    long long t_phase;      // telemetry: cycle count at the start of the phase, and the recursion wave's split of it
    long long tele[4];      // [start -> first node, first node -> last node done, last node done -> the role returns, line search inside the role]
    long long wait[4];      // cycles each wave spent at the tick barriers
};
// tick watchdog: more ticks than any phase can need means a broken protocol; every wave counts the same barriers, so all of
// them see the limit at the same tick and unwind together (no wave is left behind at a barrier)
struct Ticker {
    int n = 0, limit = 0;
    bool dead = false;
    long long wait = 0;     // telemetry: cycles spent at the tick barriers
};
// the recursion's and the gains wave's barrier: LDS traffic complete (their hand-over goes through LDS)
// (-DFUSED_TELEMETRY: cycles spent at the barriers, per wave -- two s_memtime and their waits per tick, so not in the product build)
__device__ __forceinline__ void tick_lds(Ticker &tk) {
#ifdef FUSED_TELEMETRY
    const long long t0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef FUSED_TELEMETRY
    tk.wait += (long long)__builtin_readcyclecounter() - t0;
#endif
    if (++tk.n > tk.limit) tk.dead = true;
}
// the producers' barrier: their global stores (the node derivatives) complete as well
__device__ __forceinline__ void tick_mem(Ticker &tk) {
#ifdef FUSED_TELEMETRY
    const long long t0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef FUSED_TELEMETRY
    tk.wait += (long long)__builtin_readcyclecounter() - t0;
#endif
    if (++tk.n > tk.limit) tk.dead = true;
}
__device__ __forceinline__ int lds_flag(const int &f) { return *const_cast<const volatile int *>(&f); }

// the gains wave of the two-wave backward pass
template <bool FUSED>
__device__ __forceinline__ void backward_gains_wave(BackwardLds<2> &s, double *ws, double *sc, const IkLayout &L, int T, int lane, FusedCtl *ctl,
                                                    Ticker &tk) {
    const bool row = lane < kNDX;
    double d1 = 0.0, st = 0.0;
    for (int k = 0;;) {
        int t;
        if (FUSED) {
            tick_lds(tk);
            if (tk.dead || tk.n >= lds_flag(ctl->done_tick)) return;     // (every wave leaves after this same barrier)
            if (lds_flag(ctl->hand_count) <= k) continue;                       // a tick without a hand-over (the recursion waits for a node)
            t = s.tnode[k & 1];
            if (t < 0) {    // the pass is over (-1) or was given up (-2): nothing else is handed over in this phase
                if (t == -1 && lane == kQuLane) { sc[S_D1] = d1; sc[S_STOP] = st; }
                if (lane == 0) ctl->done_tick = tk.n + 1;
                ++k;
                continue;
            }
        } else {
            bwd_barrier();
            t = s.tnode[k & 1];
            if (t < 0) break;
        }
        const int buf = k & 1;
        ++k;
        if (t == T - 1) { d1 = 0.0; st = 0.0; }      // first node of a pass (a pass that failed is started again)
        double y[kNV], idg[kNV], quv[kNV];
        {
            const double *src = row ? s.Ys[buf] + lane * LDK : s.yu[buf];
            UNROLL_RBD for (int p = 0; p < kNV; ++p) y[p] = src[p];
            UNROLL_RBD for (int p = 0; p < kNV; ++p) { idg[p] = s.idg[buf][p]; quv[p] = s.qu[buf][p]; }      // (this wave has time)
        }
        {   // back substitution L^T k = y, column by column (as in the one-wave kernel)
            const unsigned lc_addr = lds_offset(s.Lc[buf]);
            double2_t lb[18];
            int cur = -1;
            UNROLL_RBD for (int p = kNV - 1; p >= 0; --p) {
                y[p] *= idg[p];
                UNROLL_RBD for (int q = p - 1; q >= 0; --q) {
                    const int idx = p * (p - 1) / 2 + q, bb = idx / 36, e = idx % 36;
                    if (bb != cur) { lds_read_b128x18(lc_addr + (unsigned)bb * 288, lb); cur = bb; }
                    y[q] -= ((e & 1) ? lb[e >> 1].y : lb[e >> 1].x) * y[p];
                }
            }
        }
        UNROLL_RBD for (int p = 0; p < kNV; ++p) { d1 += quv[p] * y[p]; st += quv[p] * quv[p]; }   // d1 = Qu.k, stop = |Qu|^2
        if (row) {
            double *Kg = ws + L.K + (long)t * kNV * kNDX + lane;
            UNROLL_RBD for (int p = 0; p < kNV; ++p) { Kg[p * kNDX] = y[p]; asm volatile("" ::: "memory"); }
        } else if (lane == kQuLane) {
            UNROLL_RBD for (int p = 0; p < kNV; ++p) ws[L.kff + (long)t * kNV + p] = y[p];
        }
    }
    if (lane == kQuLane) { sc[S_D1] = d1; sc[S_STOP] = st; }      // (the multi-kernel path: after the end marker)
}

// The recursion's wave.  FUSED: inside ik_fused_kernel -- barriers are ticks, a node waits (whole ticks) until the producer waves
// have left its derivatives in the workspace, the total cost is summed at the end of the pass (the node costs come from LDS) and
// the wave never returns early: it leaves with every other wave of the workgroup, after the tick the gains wave names (done_tick).
template <int NWB, bool FUSED>
__device__ __forceinline__ void backward_main_wave(const IkBatchArgs &a, long b, BackwardLds<NWB> &s, int lane, FusedCtl *ctl, Ticker &tk) {
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + uni(b) * L.total;      // (scalar registers: the base of every workspace access of the pass)
    double *sc = ws + L.scal;
    const int T = a.T;
    bool feas = sc[S_FEAS] != 0.0;
    const bool wasfeas = sc[S_WASFEAS] != 0.0;
    const bool recalc = sc[S_RECALC] != 0.0;
    double xreg = sc[S_XREG];       // (with the flags: read further down it is one more memory latency in front of the first node)
    for (int i = lane; i < NWB * kPadRows * LDK; i += 64) s.Ys[0][i] = 0.0;                  // the padding of the MFMA operand image(s)
    for (int i = kNDX * LD + lane; i < kPadRows * LD + 16; i += 64) s.N[i] = 0.0;
    int hand = 0;       // NWB = 2: nodes handed to the gains wave so far (buffer = hand & 1)
    bool gave_up = false;
    // FUSED: node t may be read once its derivatives are there
    auto wait_node = [&](int t) { if (FUSED) { while (lds_flag(ctl->ready[t]) != ctl->stamp && !tk.dead) tick_lds(tk); } };
    if (FUSED) {
        while (lds_flag(ctl->state_ready) != ctl->stamp && !tk.dead) tick_lds(tk);          // the gaps need every node's Euler step
        wait_node(T);
        if (tk.dead) return;        // (the watchdog: no barrier may follow; the kernel reports and leaves)
    }
    long long t_first = 0, t_last = 0;
    if (FUSED) t_first = __builtin_readcyclecounter();

    if (recalc) {
        // SolverDDP::calcDiff tail: total cost and the gaps fs
        if (!FUSED) {
            // (one lane adding T + 1 values it loads one after the other: a global-memory latency per node, ~0.6 us each -- 18 us of
            // a 240 us pass at T = 30.  The loads go out side by side, one node per lane; the additions keep the reference's order.)
            const double c = sum_nodes_in_turn([&](int node) { return ws[L.fs + (long)node * kNDX]; }, T);
            if (lane == 0) sc[S_COST] = c;
            wave_sync();
        }
        if (!feas) {
            double mx = 0.0;
            for (int node = lane; node <= T; node += 64) {      // (one node per lane, 64 at a time)
                double d[kNDX];
                const double *xa = ws + L.xs + (long)node * kNX;
                const double *xb = node == 0 ? a.x0 + b * kNX : ws + L.xnext + (long)(node - 1) * kNX;
                state_diff_q(xa, xb, d);
                UNROLL_RBD for (int i = 0; i < kNDX; ++i) { ws[L.fs + (long)node * kNDX + i] = d[i]; mx = fmax(mx, fabs(d[i])); }
            }
            feas = !__any(!(mx < 1e-16));       // th_gaptol_
            if (lane == 0) sc[S_FEAS] = feas ? 1.0 : 0.0;
        } else if (!wasfeas) {
            for (long i = lane; i < (long)(T + 1) * kNDX; i += 64) ws[L.fs + i] = 0.0;
        } else {
            for (int node = lane; node <= T; node += 64) ws[L.fs + (long)node * kNDX] = 0.0;   // the parked node costs
        }
        wave_sync();
    }

    const bool row = lane < kNDX;                  // owns row `lane` of the 36x36 matrices
    // lanes 36..53 shadow lanes 18..35 (same reads, same arithmetic, no writes): their copy of rows 18..35 of G becomes the
    // rows of Q_uu in the elimination, where lanes 18..35 themselves carry rows of Q_xu
    // (lanes 54..63 own nothing; their addresses point at row 4, whose LDS banks no row of their lane group uses: row 0 shares
    // its banks with row 32 and cost every row / column read of the wave a conflict cycle)
    const int r = row ? lane : (lane < kNDX + kNV ? lane - kNV : 4);
    const bool ul = lane >= kNV && lane < kNDX;    // owns control q = lane - 18 (rows 18..35 are the v-rows)
    const int uq = ul ? lane - kNV : 0;
    const unsigned row_addr = lds_offset(s.N + r * LD), col_addr = lds_offset(s.N + r);
    const unsigned a6_addr = lds_offset(s.A6), b6_addr = lds_offset(s.B6), fs_addr = lds_offset(s.fs);
    const LxxOffsets lxo = lxx_offsets(lane);
    double d1, d2, st;
#ifdef BWD_PROFILE
    long long pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#endif
    for (;;) {   // computeDirection with regularisation retries (solver-ddp.cpp solve())
        bool bad = false;
        d1 = 0.0; d2 = 0.0; st = 0.0;
        double m[kNDX];   // the one 36-wide register row: V -> N -> G -> V_xx -> V
        {   // V_T = L_xx(T) + xreg I, put together in the tile layout its compact form is made for and read back by rows
            LxxLane lt;
            lxx_lane_load(ws, L, T, lane, lxo, lt);
            mfma_acc_t acc[6];
            UNROLL_RBD for (int tl = 0; tl < 6; ++tl) acc[tl] = mfma_acc_t{0.0, 0.0, 0.0, 0.0};
            lxx_add_tiles(acc, lt, lane);
            tiles_to_rows(s.N, acc, xreg, lane);
            wave_sync();
            lds_read_row36_b64(row_addr, m);
            wave_sync();
        }
        double vx = row ? ws[L.Lx + (long)T * kNDX + r] : 0.0;
        if (!feas) {
            if (row) s.fs[r] = ws[L.fs + (long)T * kNDX + r];
            wave_sync();
            double acc = 0.0, fsv[kNDX];
            lds_read_row36(fs_addr, fsv);
            UNROLL_RBD for (int j = 0; j < kNDX; ++j) acc += m[j] * fsv[j];
            vx += acc;
            wave_sync();
        }

        // What a node reads from the workspace -- its L_xx row (36 doubles per lane), L_x / L_u / L_uu entries, the Jintegrate blocks
        // and its gap -- is requested ONE NODE AHEAD: issued at the top of node t + 1, it has the whole of that node (~20K cycles)
        // to arrive.  Loaded at the top of its own node the row had ~5K cycles before its first use and the small vectors none
        // (a global-memory latency exposed twice per node: 3.4K + 2.1K of 22.5K cycles, tools/bwd_profile.py).
        double lx_t, lu_t, luu_t, a6_t, b6_t, fs_t, dt;
        const unsigned off_r = 8u * (unsigned)(row ? r : 0), off_u = 8u * (unsigned)uq, off_36 = 8u * (unsigned)(lane < 36 ? lane : 0),
                       off_uu = 8u * (unsigned)(lane >= kNDX && lane < kNDX + kNV ? lane - kNDX : 0);
        auto fetch_node = [&](int t, double &o_lx, double &o_lu, double &o_luu, double &o_a6, double &o_b6, double &o_fs, double &o_dt) {
            // (uniform base + 32-bit lane offset, every lane an existing element: unconditional loads, zero selected afterwards)
            o_dt = a.dt[b * T + t];
            const double xlx = *at_u(ws + L.Lx + (long)t * kNDX, off_r), xlu = *at_u(ws + L.Lu + (long)t * kNV, off_u);
            const double xluu = *at_u(ws + L.Luu + (long)t * kNV, off_uu);
            const double xa6 = *at_u(ws + L.A6 + (long)t * 36, off_36), xb6 = *at_u(ws + L.B6 + (long)t * 36, off_36);
            const double xfs = *at_u(ws + L.fs + (long)t * kNDX, off_r);
            o_lx = row ? xlx : 0.0;
            o_lu = ul ? xlu : 0.0;
            o_luu = lane >= kNDX && lane < kNDX + kNV ? xluu : 0.0;   // on the lane of Q_uu row p
            o_a6 = lane < 36 ? xa6 : 0.0;
            o_b6 = lane < 36 ? xb6 : 0.0;
            o_fs = row ? xfs : 0.0;
        };
        bool have = false;      // node t's small reads were requested a node ahead
        for (int t = T - 1; t >= 0; --t) {
#ifdef BWD_PROFILE
            pt0 = __builtin_readcyclecounter();
#endif
            if (!have) {        // the first node of a pass; in the fused kernel also a node the producers had not finished a tick ago
                wait_node(t);
                if (FUSED && tk.dead) return;
                fetch_node(t, lx_t, lu_t, luu_t, a6_t, b6_t, fs_t, dt);
            }
            if (lane < 36) { s.A6[lane] = a6_t; s.B6[lane] = b6_t; }
            if (row) { s.Vx[r] = vx; s.fs[r] = fs_t; }
            const double dt_t = dt;      // (this node's; the variables above are about to be overwritten by the requests for node t - 1)
            double nlx = 0.0, nlu = 0.0, nluu = 0.0, na6 = 0.0, nb6 = 0.0, nfs = 0.0, ndt = 0.0;
            // (lean build: no request a node ahead -- its fourteen registers live through the whole node; the SIMD's other wave covers the wait)
            const bool next_there = !(NWB == 1 && kBwdLean) && t > 0 && (!FUSED || lds_flag(ctl->ready[t - 1]) == ctl->stamp);
            if (next_there) fetch_node(t - 1, nlx, nlu, nluu, na6, nb6, nfs, ndt);
            // the node's compact L_xx: requested here, used at the very end of the node (the Schur update) -- the whole node to arrive
            LxxLane lxx;
            if (!(NWB == 1 && kBwdLean)) lxx_lane_load(ws, L, t, lane, lxo, lxx);
            wave_sync();
            PSTAMPV(0, m[0]) LEAN_FENCE
            apply_FxT(m, a6_addr, b6_addr, dt_t);                  // column r of N = F_x^T V
            if (row) { UNROLL_RBD for (int i = 0; i < kNDX; ++i) s.N[i * LD + r] = m[i]; }
            wave_sync();
            PSTAMPV(1, m[0]) LEAN_FENCE
            lds_read_row36_b64(row_addr, m);                         // row r of N
            apply_FxT(m, a6_addr, b6_addr, dt_t);                  // row r of G = N F_x
            PSTAMPV(2, m[35]) LEAN_FENCE
            // Q_x = L_x + F_x^T V_x ;  Q_u = L_u + dt (F_x^T V_x)[v]
            double fvx;
            if (r < 6) { fvx = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) fvx += s.A6[6 * c + r] * s.Vx[c]; }
            else if (r < kNV) fvx = s.Vx[r];
            else if (r < kNV + 6) { fvx = 0.0; UNROLL_RBD for (int c = 0; c < 6; ++c) fvx += s.B6[6 * c + (r - kNV)] * s.Vx[c]; fvx = dt_t * fvx + s.Vx[r]; }
            else fvx = dt_t * s.Vx[r - kNV] + s.Vx[r];
            const double qx = lx_t + fvx;
            const double qu = ul ? lu_t + dt_t * fvx : 0.0;   // Q_u[q] on lane 18 + q
            // The factorisation and the forward substitutions in one elimination.  Y^T = Q_xu L^-T obeys the recurrence of the
            // rows of L itself:  Y^T[r][j] = (Q_xu[r][j] - sum_{k<j} Y^T[r][k] L[j][k]) / L[j][j],  so the rows of Q_xu ride along
            // as extra rows of the matrix being factorised -- [Q_xu (lanes 0..35); Q_uu (lanes 36..53); Q_u^T (lane 54)], one row
            // per lane, the same instructions for all of them: the right-looking Cholesky loop of Q_uu leaves L on the u-lanes,
            // Y^T on the x-lanes and y_u = L^-1 Q_u on lane 54, and the 37 forward substitutions cost nothing beyond it.
            // Q_uu row p = dt Q_xu[18 + p][:] is at hand on lane 36 + p, which shadows lane 18 + p.
            double al[kNV], quv[kNV];
            const bool urow = lane >= kNDX && lane < kQuLane;
            UNROLL_RBD for (int q = 0; q < kNV; ++q) {
                const double xq = dt_t * m[kNV + q];                       // Q_xu[r][q]
                quv[q] = lane_value(qu, kNV + q);
                al[q] = urow ? dt_t * xq : (lane == kQuLane ? quv[q] : xq);   // Q_uu[p][q] without its diagonal term / Q_u / Q_xu
            }
            const double dgv = urow ? luu_t + xreg : 0.0;                // L_uu + reg of control p on lane 36 + p
            if (row) { UNROLL_RBD for (int j = 0; j < kNDX; ++j) s.N[r * LD + j] = m[j]; }       // row of G -> LDS (L_xx joins it in the tiles)
            PSTAMPV(3, al[17]) LEAN_FENCE
            // Pivots and the column entries every lane needs travel by v_readlane (wave-uniform, no LDS, no waiting).  A
            // non-positive or NaN pivot fails the pass (Eigen::LLT info != Success).
            // One reciprocal square root per pivot serves both uses (1 / piv = rs^2 for the elimination, rs for the scaling of the
            // column), and the NEXT pivot's is started as soon as its entry has had this pivot's update -- before the rest of
            // the trailing update, whose independent instructions then fill the waits of that dependent chain.
            double idg[kNV];
            double rs, mydg = 0.0;
            const int buf = NWB == 2 ? (hand & 1) : 0;      // the hand-over buffer of this node
            int badpiv = 0;      // (an integer OR per pivot: as `bad = bad || ...` hipcc kept all eighteen pivots for one late evaluation -- in scratch, in the 256-register build)
            {
                const double piv = lane_value(al[0], kNDX) + lane_value(dgv, kNDX);
                badpiv |= piv > 0.0 ? 0 : 1; asm volatile("" : "+v"(badpiv));       // (taken NOW: left to itself hipcc keeps the pivots and compares at the node's end)
                rs = rsqrt64(piv);
            }
            UNROLL_RBD for (int j = 0; j < kNV; ++j) {
                if (!(NWB == 1 && kBwdLean)) idg[j] = rs;
                if (NWB == 2 || kBwdLean) mydg = lane == kNDX + j ? rs : mydg;      // 1 / L[j][j] stays on the lane of row j (one select, no branch)
                const double f = al[j] * (rs * rs);
                double rs_next = 0.0;
                if (j + 1 < kNV) {
                    al[j + 1] -= f * lane_value(al[j], kNDX + j + 1);
                    const double piv = lane_value(al[j + 1], kNDX + j + 1) + lane_value(dgv, kNDX + j + 1);
                    badpiv |= piv > 0.0 ? 0 : 1; asm volatile("" : "+v"(badpiv));       // (taken NOW: left to itself hipcc keeps the pivots and compares at the node's end)
                    rs_next = rsqrt64(piv);
                }
                UNROLL_RBD for (int q = j + 2; q < kNV; ++q) al[q] -= f * lane_value(al[j], kNDX + q);
                al[j] *= rs;                            // L[p][j] (rows p > j) / Y^T[r][j] / y_u[j]
                rs = rs_next;
            }
            bad = bad || badpiv != 0;
            PSTAMPV(4, idg[17]) LEAN_FENCE
            // The factor goes to LDS once (packed by columns): the back substitutions read it back by broadcast -- one
            // ds_read_b128 per two entries instead of four v_readlane.
            if (urow) {      // row p of L (its p entries left of the diagonal) is at hand on lane 36 + p: packed by rows
                const int p = lane - kNDX;
                UNROLL_RBD for (int q = 0; q < kNV - 1; ++q) {
                    if (q < p) s.Lc[buf][p * (p - 1) / 2 + q] = al[q];
                }
            }
            double (&y)[kNV] = al;
            // expectedImprovement / stoppingCriteria ingredients (lane 54): d2 = -k.Quu k = -|L^T k|^2 = -|L^-1 Qu|^2; and
            // V_x = Q_x - K^T Q_u = Q_x - Y^T y_u (K = L^-T Y, y_u = L^-1 Q_u on lane 54): from the forward substitutions alone, like
            // V_xx -- the recursion does not wait for the gains
            auto improvement_and_vx = [&]() {
                UNROLL_RBD for (int p = 0; p < kNV; ++p) d2 -= y[p] * y[p];
                double w = qx;
                UNROLL_RBD for (int p = 0; p < kNV; ++p) w -= y[p] * lane_value(y[p], kQuLane);
                vx = w;
            };
            // Y = L^-1 Q_ux (column j on lane j) is all the Riccati recursion needs: Q_xu K = Q_xu Q_uu^-1 Q_ux = Y^T Y.  It goes to
            // LDS as the one operand image of the Schur update; the gains K = L^-T Y (back substitution) are for the forward pass,
            // not for the next node.  The MFMAs are issued after the substitution (their accumulators would not fit beside its
            // registers) and before the stores of K, V_x and the improvement terms, which the vector pipe does while the matrix
            // pipe works through its tiles.
            if (NWB == 2) {     // ... and y_u, the reciprocal pivots and Q_u for the gains wave, in the same instructions where possible
                double *dst = row ? s.Ys[buf] + r * LDK : s.yu[buf];
                if (row || lane == kQuLane) { UNROLL_RBD for (int p = 0; p < kNV; ++p) dst[p] = y[p]; }
                if (urow) s.idg[buf][lane - kNDX] = mydg;
                if (ul) s.qu[buf][uq] = qu;
                if (lane == 0) { s.tnode[buf] = t; if (FUSED) ctl->hand_count = hand + 1; }
            } else {
                if (row) { UNROLL_RBD for (int p = 0; p < kNV; ++p) s.Ys[buf][r * LDK + p] = y[p]; }
                if (kBwdLean) {      // the reciprocal pivots and Q_u wait in LDS (as for the gains wave of the two-wave kernel), not in registers
                    if (urow) s.idg[0][lane - kNDX] = mydg;
                    if (ul) s.qu[0][uq] = qu;
                }
                improvement_and_vx();
            }
            wave_sync();
            // V_xx = Q_xx - Y^T Y on the matrix pipe (layouts at BackwardLds): fp64 MFMA has the vector FMA rate on gfx950, so this
            // is not about flops -- one MFMA replaces 16 wave-wide FMAs in the issue stream, and every lane feeds ONE element of
            // Y per step instead of all lanes reading a whole 36 x 18 matrix by broadcast (51 ds_read_b64 per lane against 342
            // ds_read_b128 of the vector version).  The product is symmetric and V_xx is symmetrised right after: only the six
            // tiles on and above the block diagonal are computed (independent accumulators, k outermost); an off-diagonal tile is
            // stored a second time, transposed, where its mirror image belongs.
            if (NWB == 1) {   // back substitution L^T k = y, column by column: once k_p is final its multiples leave all earlier equations --
                // independent updates (the row form accumulated each k_p through a chain of dependent FMAs)
                const unsigned lc_addr = lds_offset(s.Lc[0]);
                if (kBwdLean) {
                    __builtin_amdgcn_sched_barrier(0);
                    lxx_lane_load(ws, L, t, lane, lxo, lxx);      // (lean: requested here, a back substitution ahead of the Schur update)
                    double2_t lb[9];                        // the factor in batches of 18 entries
                    int cur = -1;
                    {
                        double2_t dg[9];
                        lds_read_b128x9(lds_offset(s.idg[0]), dg);
                        UNROLL_RBD for (int p = 0; p < kNV; ++p) idg[p] = (p & 1) ? dg[p >> 1].y : dg[p >> 1].x;
                    }
                    UNROLL_RBD for (int p = kNV - 1; p >= 0; --p) {
                        y[p] *= idg[p];
                        UNROLL_RBD for (int q = p - 1; q >= 0; --q) {
                            const int idx = p * (p - 1) / 2 + q, bb = idx / 18, e = idx % 18;
                            if (bb != cur) { lds_read_b128x9(lc_addr + (unsigned)bb * 144, lb); cur = bb; }
                            y[q] -= ((e & 1) ? lb[e >> 1].y : lb[e >> 1].x) * y[p];
                        }
                    }
                } else {
                double2_t lb[18];
                int cur = -1;
                UNROLL_RBD for (int p = kNV - 1; p >= 0; --p) {
                    y[p] *= idg[p];
                    UNROLL_RBD for (int q = p - 1; q >= 0; --q) {      // descending, so the batches are met from the last to the first
                        const int idx = p * (p - 1) / 2 + q, bb = idx / 36, e = idx % 36;
                        if (bb != cur) { lds_read_b128x18(lc_addr + (unsigned)bb * 288, lb); cur = bb; }
                        y[q] -= ((e & 1) ? lb[e >> 1].y : lb[e >> 1].x) * y[p];
                    }
                }
                }
                // The gains leave the wave HERE, before the Schur update: this kernel runs two waves per SIMD (256 registers), the
                // other wave fills the matrix pipe's shadow, and y / Q_u are dead by the time the six accumulator tiles are live.
                if (kBwdLean) {
                    double2_t qv[9];
                    lds_read_b128x9(lds_offset(s.qu[0]), qv);
                    UNROLL_RBD for (int p = 0; p < kNV; ++p) quv[p] = (p & 1) ? qv[p >> 1].y : qv[p >> 1].x;
                }
                UNROLL_RBD for (int p = 0; p < kNV; ++p) { d1 += quv[p] * y[p]; st += quv[p] * quv[p]; }   // d1 = Qu.k, stop = |Qu|^2
                if (row) {      // (the empty asm keeps the stores in program order: clustered, hipcc spills a hundred registers around them)
                    double *Kg = at_u(ws + L.K + (long)t * kNV * kNDX, off_r);
                    UNROLL_RBD for (int p = 0; p < kNV; ++p) { Kg[p * kNDX] = y[p]; asm volatile("" ::: "memory"); }
                } else if (lane == kQuLane) {
                    UNROLL_RBD for (int p = 0; p < kNV; ++p) ws[L.kff + (long)t * kNV + p] = y[p];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PSTAMPV(5, vx) LEAN_FENCE
            const int li = lane & 15, lk = lane >> 4;
            mfma_acc_t acc[6];
            {
                double bv[15], c[24];
                lds_read_mfma_operand(lds_offset(s.Ys[buf] + li * LDK + lk), bv);
                lds_read_mfma_acc_upper(lds_offset(s.N + lk * LD + li), c);
                UNROLL_RBD for (int tl = 0; tl < 6; ++tl)
                    UNROLL_RBD for (int v = 0; v < 4; ++v) acc[tl][v] = c[4 * tl + v];
                lxx_add_tiles(acc, lxx, lane);                 // Q_xx = G + L_xx
                UNROLL_RBD for (int ks = 0; ks < 5; ++ks)
                    UNROLL_RBD for (int I = 0; I < 3; ++I)
                        UNROLL_RBD for (int J = I; J < 3; ++J) {
                            const int tl = (I == 0 ? 0 : I == 1 ? 2 : 3) + J;
                            acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bv[5 * I + ks], bv[5 * J + ks], acc[tl], 0, 0, 0);
                        }
            }
            if (NWB == 2) {     // while the matrix pipe works through the tiles; then the node is the gains wave's
                improvement_and_vx();
                if (FUSED) tick_lds(tk); else bwd_barrier();
                ++hand;
            }
            PSTAMPV(6, vx) LEAN_FENCE
            tiles_to_rows(s.N, acc, xreg, lane);
            wave_sync();
            PSTAMPV(7, vx) LEAN_FENCE
            lds_read_row36_b64(row_addr, m);
            {
                double col[kNDX], chk = 0.0;
                lds_read_col36_ld37(col_addr, col);
                UNROLL_RBD for (int j = 0; j < kNDX; ++j) {
                    m[j] = 0.5 * (m[j] + col[j]);
                    chk = fma(m[j], 0.0, chk);          // NaN as soon as one entry is NaN or infinite
                }
                bad = bad || !(chk == 0.0);
            }
            if (!feas) {
                double acc = 0.0, fsv[kNDX];
                lds_read_row36(fs_addr, fsv);
                UNROLL_RBD for (int j = 0; j < kNDX; ++j) acc += m[j] * fsv[j];
                vx += acc;
            }
            bad = bad || !(fabs(vx) < INFINITY);       // raiseIfNaN on Vx / Vxx
            bad = __any(bad && (row || lane < kNV));
            wave_sync();
            PSTAMPV(8, vx) LEAN_FENCE
            have = next_there;
            if (next_there) {     // the node requested at the top becomes the current one
                lx_t = nlx; lu_t = nlu; luu_t = nluu; a6_t = na6; b6_t = nb6; fs_t = nfs; dt = ndt;
            }
            if (bad || (FUSED && tk.dead)) break;
        }
        if (!bad || (FUSED && tk.dead)) break;
        // increaseRegularization; give up at reg_max (solve() returns false)
        xreg = fmin(xreg * 10.0, 1e9);
        if (lane == 0) { sc[S_XREG] = xreg; sc[S_RECALC] = 0.0; }
        if (xreg == 1e9) {
            if (lane == 0) { sc[S_DONE] = 1.0; sc[S_STATUS] = 2.0; atomicSub(a.active, 1); }
            if (FUSED) { gave_up = true; break; }
            if (NWB == 2) { if (lane == 0) s.tnode[hand & 1] = -1; bwd_barrier(); }
            return;
        }
    }
#ifdef BWD_PROFILE
    if (lane == 0) { for (int k = 0; k < 9; ++k) ws[L.Quuk + k] = (double)pc[k]; }   // the Quuk slot is unused by the solver
#endif
    if (FUSED) {
        if (tk.dead) return;
        t_last = __builtin_readcyclecounter();
        // the end marker (-1: the gains wave writes d1 and the stopping criterion; -2: the pass was given up), then the total cost
        // in SolverDDP's order from the node costs the producers left in LDS (every node has been produced by now: node 0 was
        // waited for), then ticks until the gains wave has declared the phase over
        if (lane == 0) { s.tnode[hand & 1] = gave_up ? -2 : -1; ctl->hand_count = hand + 1; }
        tick_lds(tk);
        if (tk.dead) return;
        if (!gave_up && lane == kQuLane) sc[S_D2] = d2;
        if (recalc && !gave_up) {
            const double c = sum_nodes_in_turn([&](int node) { return ctl->node_cost[node]; }, T);      // (the fused kernel: T + 1 <= 64)
            if (lane == 0) sc[S_COST] = c;
        }
        for (;;) {
            tick_lds(tk);
            if (tk.dead || tk.n >= lds_flag(ctl->done_tick)) break;
        }
        if (lane == 0) {
            ctl->tele[0] += t_first - ctl->t_phase; ctl->tele[1] += t_last - t_first; ctl->tele[2] += (long long)__builtin_readcyclecounter() - t_last;
        }
    } else if (NWB == 2) {     // the pass is over: the gains wave writes d1 and the stopping criterion on its way out
        if (lane == 0) s.tnode[hand & 1] = -1;
        bwd_barrier();
        if (lane == kQuLane) sc[S_D2] = d2;
    } else if (lane == kQuLane) { sc[S_D1] = d1; sc[S_D2] = d2; sc[S_STOP] = st; }   // the lane of the feed-forward terms
}


template <int NWB>
__global__ __launch_bounds__(64 * NWB, NWB == 1 ? BWD_WPE : 1) void ik_backward_kernel(const IkBatchArgs a) {
    __shared__ BackwardLds<NWB> s;
    const long b = slot_problem(a, blockIdx.x);
    if (b < 0) return;
    const int lane = threadIdx.x & 63;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    if (ws[L.scal + S_DONE] != 0.0) return;
    Ticker tk;
    if (NWB == 2 && threadIdx.x >= 64) { backward_gains_wave<false>(reinterpret_cast<BackwardLds<2> &>(s), ws, ws + L.scal, L, a.T, lane, nullptr, tk); return; }
    backward_main_wave<NWB, false>(a, b, s, lane, nullptr, tk);
}

// ---------------------------------------------------------------------------- forward ---
// Line search + rollout.  FOUR problems per wave, 16 lanes each: the rollout of one problem is a serial chain whose
// pieces run on a handful of lanes (one lane for the SE(3) difference, 16 for the feedback rows, five for the legs
// and the base, one for the state cost, one for the control cost and the Euler step), and pieces with different code
// cannot overlap inside a wave -- so a wave carries four independent chains through the same instruction stream.
constexpr int kFwdSub = 4, kFwdLanes = 64 / kFwdSub;
static_assert(kTrySlots >= 3 * kFwdSub - 2, "one trial slot per step length in the all-step-lengths mapping");

// Every vector a lane reads as a whole is 16-byte aligned and padded to the length of the batch reader that fetches it
// (lds_batch.h): read element by element, each LDS read waits out its own latency in front of its first use.
struct alignas(16) FwdSub {
    double dx[kNDX], u[22], x[2][40]; double part[2][kLegs + 1][10 + 3 * kFrameSlots]; double bc[3][4], ft[kFrameSlots];      // x, part: [node parity]; bc: [node mod 3]; ft: the frame terms of the node being summed
    // copies of what every node reads from HBM: per problem (regularisation reference and weights), per node (task block,
    // dt, the nominal state the feedback is taken around) -- the per-node ones are fetched one node ahead
    double xreg[40], sw[kNDX], cw[22], tk[3][kNodeTaskDoubles + 3], xs[40];     // tk: [node mod 3]
};
static_assert((10 + 3 * kFrameSlots) % 2 == 0 && kNodeTaskDoubles % 2 == 1, "FwdSub members stay 16-byte aligned");
__device__ __forceinline__ void lds_read_vec40(const double *p, double (&o)[40]) {
    double2_t t[20];
    lds_read_b128x20(lds_offset(p), t);
    UNROLL_RBD for (int i = 0; i < 20; ++i) { o[2 * i] = t[i].x; o[2 * i + 1] = t[i].y; }
}
__device__ __forceinline__ void lds_read_vec36(const double *p, double (&o)[36]) {
    double2_t t[18];
    lds_read_b128x18(lds_offset(p), t);
    UNROLL_RBD for (int i = 0; i < 18; ++i) { o[2 * i] = t[i].x; o[2 * i + 1] = t[i].y; }
}
__device__ __forceinline__ void lds_read_vec22(const double *p, double (&o)[22]) {
    double2_t t[11];
    lds_read_b128x11(lds_offset(p), t);
    UNROLL_RBD for (int i = 0; i < 11; ++i) { o[2 * i] = t[i].x; o[2 * i + 1] = t[i].y; }
}
struct ForwardLds { RobotModelDev m; FwdSub sub[kFwdSub]; double vote[kFwdSub], ctry[kFwdSub]; };

// Two mappings of the four sub-groups of a wave (a.fwd_spec, chosen by the host per DDP iteration):
//   0  four PROBLEMS per wave, each trying its step lengths 2^-n one after the other (many active problems);
//   1  four STEP LENGTHS of one problem per wave (2^-(4k+s) in round k): SolverDDP tries them in order and takes the
//      first that passes, trials are independent of each other, so running them side by side and taking the first
//      passing one is the same decision -- it trades idle SIMDs for a 4x shorter serial chain when few problems
//      are still iterating.
//
// NW > 1 (used with the speculative mapping): wave 1 takes the leg walks of every node (with NW = 2 also the sum of their
// parts; from three waves on that sum runs one node behind on wave 2 / 3, see forward_roles), wave 2 the state
// regularisation residual (an SE(3) difference, as long as a leg walk; with NW = 2 it stays on wave 0, where it is the
// SAME code as the chain's dx = xs (-) x with other operands and lane 5 runs it in the instruction stream lane 0 needs
// anyway), while wave 0 runs the chain x -> dx -> u -> Euler step -> next x; they meet ONCE per node (one barrier: what the
// next node reads is written into rotating LDS slots before it).  Different code cannot overlap inside a wave, but it can
// across the waves of a workgroup.  Two waves while the problems still cover the chip (<= 1024 active), three once a third
// of the SIMDs would be idle anyway.
// Barrier of the forward pass' workgroup.  Its waves exchange through LDS only (the trial trajectories they write to the
// workspace are read back after the kernel, or by the same wave), so the barrier needs the LDS operations complete, not the
// global ones: __syncthreads() would also drain the rows of K and the task blocks that are fetched one node ahead, five times
// per node.  One wave (the many-problems mapping): no s_barrier at all (wave_sync above).
template <int NW>
__device__ __forceinline__ void fwd_sync() {
    if (NW == 1) wave_sync();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// FUSED (ik_fused_kernel: NW = 4, wave 3 takes no part but keeps the barriers): the problem is given, its four step lengths run
// side by side (further rounds of four in the same call), nothing is appended to the active list, and no wave returns before
// the last barrier.
// ROLES (bit 0: the chain, bit 1: the robot walks and the residual costs, bit 2: the state-regularisation residual): what THIS wave
// does, a compile-time constant.  The workgroup's waves run different instantiations of this one body (ik_forward_kernel below; in
// the fused kernel the chain role inline, the others as non-inlined functions): as run-time conditions inside one instantiation
// the roles' register arrays interfered in hipcc's allocator -- 512 registers and 124 spilled ones for code that needs half of that.
// With three waves or more the robot walk of a node and the sum of its parts (totals, residuals, costs: one lane's serial work,
// about as long as the walk) are on DIFFERENT waves, the sum one node behind the walk: the walk wave was the one the others waited
// for at every node (10.5K cycles against the chain's 8.4K, tools/bwd_profile.py), now each half is shorter than the chain.  What
// a node leaves for the node after it -- the parts, the cost terms, the task block -- is therefore kept per node PARITY in LDS,
// and the node costs enter the trial's cost one node late (same order, same sum).
constexpr int kRoleChain = 1, kRoleCost = 2, kRoleReg = 4, kRoleSum = 8;
constexpr int forward_roles(int NW, int wave) {
    return NW == 1 ? 15
         : NW == 2 ? (wave == 0 ? kRoleChain | kRoleReg : kRoleCost | kRoleSum)
         : NW == 3 ? (wave == 0 ? kRoleChain : wave == 1 ? kRoleCost : kRoleReg | kRoleSum)
                   : (wave == 0 ? kRoleChain : wave == 1 ? kRoleCost : wave == 2 ? kRoleReg : kRoleSum);
}
template <int NW, bool FUSED, int ROLES>
__device__ __forceinline__ void forward_body(const IkBatchArgs &a, ForwardLds &s, long b_fused) {
    const int lane = threadIdx.x & 63, wave = NW > 1 ? (int)(threadIdx.x >> 6) : 0, si = lane / kFwdLanes, l = lane % kFwdLanes;
    constexpr bool do_chain = (ROLES & kRoleChain) != 0, do_cost = (ROLES & kRoleCost) != 0, do_reg = (ROLES & kRoleReg) != 0,
                   do_sum = (ROLES & kRoleSum) != 0;
    constexpr bool kSplit = NW >= 3;       // the sum of a node's parts runs on another wave than its walk, one node behind
    const bool spec = FUSED || a.fwd_spec != 0;
    // fwd_spec == 4: THREE workgroups per problem, workgroup g trying step lengths 2^-(4g + s): all ten in one round, on
    // separate CUs (inside one workgroup the register budget of seven waves did not allow it, EXPERIMENTS.md 9); the last of the
    // three to finish takes SolverDDP's decision for the problem
    const bool all10_batch = !FUSED && NW == 3 && a.fwd_spec == 4;
    // ... or for the flagged problems only (S_WIDENOW, set by the pass before): their two extra workgroups are the FIRST
    // 2 kWideMax of the grid, two per place of the wide list (first, so that they start with the first wave of workgroups
    // when the grid is larger than the chip; the unused ones return at once), the regular ones follow
    const bool widegrid = !FUSED && NW > 1 && !all10_batch && a.wide != nullptr;
    const long blk = (long)blockIdx.x - (widegrid ? 2 * kWideMax : 0);
    const bool extra = widegrid && blk < 0;
    int grp = all10_batch ? (int)(blockIdx.x % 3) : 0;
    long b;
    if (FUSED) b = b_fused;
    else if (extra) {
        const int e = (int)(blockIdx.x >> 1), cur = a.iter & 1, nw = a.wcount[cur] < kWideMax ? a.wcount[cur] : kWideMax;
        b = e < nw ? (long)a.wide[cur * kWideMax + e] : -1;
        if (b >= a.B || b < -1) { index_error(a, IK_ERR_WIDE_ENTRY, (int)b); b = -1; }
        grp = 1 + (int)(blockIdx.x & 1);
    } else b = slot_problem(a, all10_batch ? blk / 3 : spec ? blk : blk * kFwdSub + si);
    const bool pvalid = b >= 0;
    const long bb = pvalid ? b : 0;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + bb * L.total;
    double *sc = ws + L.scal;
    const bool all10 = all10_batch || (!FUSED && NW > 1 && spec && a.wide && pvalid && sc[S_WIDENOW] != 0.0);
    FwdSub &q = s.sub[si];
    bool live = pvalid && sc[S_DONE] == 0.0;     // this sub-group still has a line search to do
    // (the problem's scalars in one go with the flag: a load behind the barrier below would be one more exposed memory latency)
    const double cost = sc[S_COST], d1 = sc[S_D1], d2 = sc[S_D2];
    const bool feas = sc[S_FEAS] != 0.0;
    if (!FUSED && !__any(live)) return;
    const int T = a.T, nn = a.T + 1;
    const int tslot = spec ? 4 * grp + si : 0;       // where this sub-group's trial trajectory goes
    const long xs_try = L.xs_try + (long)tslot * nn * kNX, us_try = L.us_try + (long)tslot * T * kNV;
    if (!FUSED) {   // the robot model is read many times per node: stage it in LDS once (the fused kernel did, when it started)
        stage_model<64 * NW>(a.model, &s.m);
    }
    const double *gsw = batch_ptr(a.state_w, a.s_state_w, bb), *gcw = batch_ptr(a.ctrl_w, a.s_ctrl_w, bb), *gxr = a.x_reg + bb * a.s_x_reg;
    const bool per_node = (a.sn_state_w | a.sn_x_reg | a.sn_ctrl_w) != 0;    // time-varying regularisation (acyclic plans)
    auto stage_reg = [&](int t) {     // regularisation reference and weights of node t into LDS
        for (int i = l; i < kNX; i += kFwdLanes) q.xreg[i] = gxr[a.sn_x_reg * t + i];
        for (int i = l; i < kNDX; i += kFwdLanes) q.sw[i] = gsw[a.sn_state_w * t + i];
        if (t < a.T || a.sn_ctrl_w == 0) { for (int i = l; i < kNV; i += kFwdLanes) q.cw[i] = gcw[a.sn_ctrl_w * t + i]; }
    };
    if (live && do_chain) {      // node 0's: all of a lane's loads in flight together (stage_reg's loops wait for each load in turn: eight round trips in front of the first node)
        static_assert(kNX <= 3 * kFwdLanes && kNDX <= 3 * kFwdLanes && kNV <= 2 * kFwdLanes, "up to three elements per lane and array");
        double vx[3], vs[3], vc[2];
        UNROLL_RBD for (int k = 0; k < 3; ++k) { const int i = l + kFwdLanes * k; vx[k] = gxr[i < kNX ? i : kNX - 1]; vs[k] = gsw[i < kNDX ? i : kNDX - 1]; }
        UNROLL_RBD for (int k = 0; k < 2; ++k) { const int i = l + kFwdLanes * k; vc[k] = gcw[i < kNV ? i : kNV - 1]; }
        UNROLL_RBD for (int k = 0; k < 3; ++k) { const int i = l + kFwdLanes * k; if (i < kNX) q.xreg[i] = vx[k]; if (i < kNDX) q.sw[i] = vs[k]; }
        UNROLL_RBD for (int k = 0; k < 2; ++k) { const int i = l + kFwdLanes * k; if (i < kNV) q.cw[i] = vc[k]; }
    }
    fwd_sync<NW>();
    const RobotModelDev &m = s.m;
    const double *state_w = q.sw, *ctrl_w = q.cw, *x_reg = q.xreg;
    const double *gtasks = a.tasks + bb * nn * kNodeTaskDoubles, *gdt = a.dt + bb * T;
    const bool owner = live;            // sub-groups that take part at all
    bool accepted = false, widen = false;
    int win = 0;                        // trial slot holding the accepted trajectory
    double alpha = 1.0, cost_try = 0.0;
#ifdef BWD_PROFILE
    long long fwork = 0, fwait = 0, fph[5] = {0, 0, 0, 0, 0}, fpt = 0, fsum = 0;
#define FSTAMP(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long now_ = __builtin_readcyclecounter(); fph[k] += now_ - fpt; fpt = now_; }
#else
#define FSTAMP(k)
#endif
    for (int round = 0; round < 10; ++round) {   // alphas_ = 2^-n, n = 0..9
        if (!__any(live)) break;
        const int ia = all10 ? 4 * grp + si : spec ? kFwdSub * round + si : round;
        bool run = live && ia < 10;     // false once this trial has failed (tryStep threw) or there is no step length left
        const double al = ldexp(1.0, -ia);
        if (!spec && live) alpha = al;
        double ctry = 0.0;
        if (run && do_chain) {
            if (per_node && round > 0) stage_reg(0);
            double v0[3], v1[3], v2[3];      // x0 (it sits in trial slot 0), the nominal x0, node 0's task block with its dt behind it
            UNROLL_RBD for (int k = 0; k < 3; ++k) {
                const int i = l + kFwdLanes * k, ix = i < kNX ? i : kNX - 1;
                v0[k] = ws[L.xs_try + ix]; v1[k] = ws[L.xs + ix];
                v2[k] = i < kNodeTaskDoubles ? gtasks[i] : gdt[0];
            }
            UNROLL_RBD for (int k = 0; k < 3; ++k) {
                const int i = l + kFwdLanes * k;
                if (i < kNX) { q.x[0][i] = v0[k]; q.xs[i] = v1[k]; }
                if (i <= kNodeTaskDoubles) q.tk[0][i] = v2[k];
            }
        }
        // feedback rows of node 0 (lanes 0..8 own rows l and l + 9): fetched one node ahead of their use from here on
        double kp0[kNDX], kp1[kNDX], up0 = 0.0, up1 = 0.0, fp0 = 0.0, fp1 = 0.0;
        if (run && do_chain && l < 9) {
            const double *K0 = ws + L.K + (long)l * kNDX, *K1 = K0 + 9 * kNDX;
            UNROLL_RBD for (int j = 0; j < kNDX; ++j) { kp0[j] = K0[j]; kp1[j] = K1[j]; }
            up0 = ws[L.us + l]; up1 = ws[L.us + l + 9]; fp0 = ws[L.kff + l]; fp1 = ws[L.kff + l + 9];
        }
        fwd_sync<NW>();
        double dt_behind = 0.0;          // kSplit: the time step of the node whose cost is taken in at this node
        for (int t = 0; t <= T + (kSplit ? 1 : 0); ++t) {   // t == T: terminal node (cost only); kSplit: one more turn for its sum
            if (!__any(run)) break;
            const bool body = t <= T;           // a node to evaluate (false in the extra turn)
            const int p = t & 1, ts = kSplit ? t - 1 : t, ps_ = ts & 1;       // ts: the node whose parts are summed / whose cost is taken in
            const int m3 = t % 3, m3n = (t + 1) % 3, m3s = kSplit ? (t + 2) % 3 : m3;     // this node's, the next node's, node ts's slot of three
#ifdef BWD_PROFILE
            const long long fnode0 = __builtin_readcyclecounter();
            fpt = fnode0;
#endif
            const bool terminal = t >= T;
            NodeTasks tk{q.tk[m3]};
            const double dtn = terminal ? 0.0 : q.tk[m3][kNodeTaskDoubles];
            // next node's task block / dt / nominal state: requested now, parked in LDS at the end of this node
            double ntk[3] = {0.0, 0.0, 0.0}, nxs[3] = {0.0, 0.0, 0.0};
            if (run && do_chain && !terminal) {
                const double *g = gtasks + (long)(t + 1) * kNodeTaskDoubles;
                UNROLL_RBD for (int k = 0; k < 3; ++k) {
                    const int i = l + kFwdLanes * k;
                    if (i < kNodeTaskDoubles) ntk[k] = g[i];
                    else if (i == kNodeTaskDoubles) ntk[k] = t + 1 < T ? gdt[t + 1] : 0.0;
                    if (i < kNX) nxs[k] = ws[L.xs + (long)(t + 1) * kNX + i];
                }
            }
            // phase 1: dx = xs[t] (-) x on lane 0 (feeds the feedback) and the state regularisation residual x_reg (-) x on
            // lane 5 (feeds the cost) -- one instruction stream when one wave does both
            const bool want_dx = do_chain && l == 0 && !terminal, want_rs = do_reg && l == 5 && body && tk.state_w() != 0.0;
            if (run && (want_dx || want_rs)) {
                double d[kNDX], xa[40], xb[40];
                lds_read_vec40(want_dx ? q.xs : x_reg, xa);
                lds_read_vec40(q.x[p], xb);
                state_diff_q(xa, xb, d);
                if (want_dx) { UNROLL_RBD for (int i = 0; i < kNDX; ++i) q.dx[i] = d[i]; }
                else {
                    double acc = 0.0, swv[kNDX];
                    lds_read_vec36(state_w, swv);
                    UNROLL_RBD for (int i = 0; i < kNDX; ++i) acc += swv[i] * d[i] * d[i];
                    q.bc[m3][2] = tk.state_w() * 0.5 * acc;
                }
            } else if (run && do_reg && l == 5 && body) q.bc[m3][2] = 0.0;
            FSTAMP(0)
            // phase 2 (needs x only): the robot walk.  With a wave of its own (NW > 1) on all 16 lanes of the sub-group -- lane
            // 4 leg + j takes joint j of the leg, lane 3 the base body (rbd_quad.h::quad_part16); on the shared wave of the
            // many-problems mapping legs on lanes 0..3, base body on lane 4
            if (NW > 1) {
                if (run && do_cost && body) {
                    int fid[kFrameSlots];
                    UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) fid[f] = tk.frame_w(f) != 0.0 ? tk.frame_id(f) : -1;
                    PartSum ps;
                    quad_part16(m, q.x[p], fid, l, ps);
                    const int row = l == 3 ? kLegs : ((l & 3) == 0 ? (l >> 2) : -1);
                    if (row >= 0) {
                        q.part[p][row][0] = ps.mass;
                        UNROLL_RBD for (int c = 0; c < 3; ++c) q.part[p][row][1 + c] = ps.h1[c];
                        UNROLL_RBD for (int c = 0; c < 6; ++c) q.part[p][row][4 + c] = ps.hO[c];
                        UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
                            UNROLL_RBD for (int c = 0; c < 3; ++c) q.part[p][row][10 + 3 * f + c] = ps.fhit[f] ? ps.fx[f][c] : 0.0;
                    }
                }
            } else if (run && do_cost && l <= kLegs) {
                int fid[kFrameSlots];
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) fid[f] = tk.frame_w(f) != 0.0 ? tk.frame_id(f) : -1;
                PartSum ps;
                double xv[40], qj[kLegJoints], vj[kLegJoints];
                lds_read_vec40(q.x[p], xv);
                UNROLL_RBD for (int j = 0; j < kLegJoints; ++j) {      // this lane's leg out of the four, by selects (no indexed registers)
                    qj[j] = l == 0 ? xv[7 + j] : l == 1 ? xv[10 + j] : l == 2 ? xv[13 + j] : xv[16 + j];
                    vj[j] = l == 0 ? xv[kNQ + 6 + j] : l == 1 ? xv[kNQ + 9 + j] : l == 2 ? xv[kNQ + 12 + j] : xv[kNQ + 15 + j];
                }
                quad_part(m, xv, qj, vj, fid, l, ps);
                q.part[p][l][0] = ps.mass;
                UNROLL_RBD for (int c = 0; c < 3; ++c) q.part[p][l][1 + c] = ps.h1[c];
                UNROLL_RBD for (int c = 0; c < 6; ++c) q.part[p][l][4 + c] = ps.hO[c];
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f)
                    UNROLL_RBD for (int c = 0; c < 3; ++c) q.part[p][l][10 + 3 * f + c] = ps.fhit[f] ? ps.fx[f][c] : 0.0;
            }
            FSTAMP(1)
            if (NW == 1) fwd_sync<NW>();   // with several waves each side hands over inside its own wave (LDS keeps a wave's order)
            // phase 3: feedback u = u - alpha k - K dx, two rows per lane (0..8 own rows l and l + 9), rows fetched a node ahead
            if (run && do_chain && !terminal && l < 9) {
                double v0 = up0 - al * fp0, v1 = up1 - al * fp1, dxv[kNDX];
                lds_read_vec36(q.dx, dxv);
                UNROLL_RBD for (int j = 0; j < kNDX; ++j) { v0 -= kp0[j] * dxv[j]; v1 -= kp1[j] * dxv[j]; }
                q.u[l] = v0; q.u[l + 9] = v1;
                ws[us_try + (long)t * kNV + l] = v0; ws[us_try + (long)t * kNV + l + 9] = v1;
                if (t + 1 < T) {   // rows of the next node travel while this node is evaluated
                    const double *K0 = ws + L.K + (long)(t + 1) * kNV * kNDX + (long)l * kNDX, *K1 = K0 + 9 * kNDX;
                    UNROLL_RBD for (int j = 0; j < kNDX; ++j) { kp0[j] = K0[j]; kp1[j] = K1[j]; }
                    up0 = ws[L.us + (long)(t + 1) * kNV + l]; up1 = ws[L.us + (long)(t + 1) * kNV + l + 9];
                    fp0 = ws[L.kff + (long)(t + 1) * kNV + l]; fp1 = ws[L.kff + (long)(t + 1) * kNV + l + 9];
                }
            }
            FSTAMP(2)
            if (NW == 1) fwd_sync<NW>();
            // phase 4: control cost + Euler step (lane 6)
            if (run && do_chain && l == 6 && body) {
                double acc = 0.0;
                if (!terminal) {
                    double cwv[22], uv[22], xv[40];
                    lds_read_vec22(ctrl_w, cwv);
                    lds_read_vec22(q.u, uv);
                    lds_read_vec40(q.x[p], xv);
                    UNROLL_RBD for (int i = 0; i < kNV; ++i) acc += cwv[i] * uv[i] * uv[i];
                    double xn[kNX];
                    euler_step<false>(xv, uv, dtn, xn, nullptr, nullptr);
                    bool bad = false;
                    UNROLL_RBD for (int i = 0; i < kNX; ++i) { ws[xs_try + (long)(t + 1) * kNX + i] = xn[i]; q.x[p ^ 1][i] = xn[i]; bad = bad || !(fabs(xn[i]) < INFINITY); }
                    q.bc[m3][1] = bad ? 1.0 : 0.0;
                } else q.bc[m3][1] = 0.0;
                q.bc[m3][3] = tk.ctrl_w() * 0.5 * acc;
            }
            FSTAMP(3)
            if (NW == 1) fwd_sync<NW>();
            // phase 5: the parts added up: CoM, centroidal momentum, their residual costs (without the state / control terms)
#ifdef BWD_PROFILE
            const long long fsum0 = __builtin_readcyclecounter();
#endif
            // phase 5: the parts added up.  Lanes 1..4 first, one frame each (its position from the five parts, its residual cost,
            // its squared residual left in LDS); then lane 0: total mass, first moment, momentum -> CoM and centroidal momentum residual costs, and the
            // node's residual cost in the order SolverDDP adds it (momentum, CoM, frames 0..3).  One lane doing all of it was the
            // longest single-lane stretch of the robot-walk wave.
            constexpr int kPw = 10 + 3 * kFrameSlots;      // 22 doubles per part
            static_assert(kPw == 22 && kLegs + 1 == 5, "the batched part readers are written for five records of 22 doubles");
            const bool sum_now = run && do_sum && ts >= 0 && ts <= T;
            if (sum_now && l >= 1 && l <= kFrameSlots) {
                NodeTasks tk{q.tk[m3s]};         // (node ts's task block: this node's, or with kSplit the one before)
                const int f = l - 1;
                const double w = tk.frame_w(f);
                double v[15], fx[3] = {0.0, 0.0, 0.0};
                lds_read_parts_triple(lds_offset(&q.part[ps_][0][10 + 3 * f]), v);
                UNROLL_RBD for (int pa = 0; pa <= kLegs; ++pa)
                    UNROLL_RBD for (int c = 0; c < 3; ++c) fx[c] += v[3 * pa + c];
                double acc = 0.0;
                UNROLL_RBD for (int k = 0; k < 3; ++k) { const double r = w != 0.0 ? fx[k] - tk.frame_ref(f)[k] : 0.0; acc += r * r; }
                q.ft[f] = acc;
            }
            if (sum_now && l == 0) {
                NodeTasks tk{q.tk[m3s]};
                double M = 0.0, h1[3] = {0, 0, 0}, hO[6] = {0, 0, 0, 0, 0, 0};
                double2_t hd[25];
                lds_read_parts_head(lds_offset(&q.part[ps_][0][0]), hd);
                auto part_at = [&](int pa, int k) -> double { return (k & 1) ? hd[5 * pa + (k >> 1)].y : hd[5 * pa + (k >> 1)].x; };
                UNROLL_RBD for (int pa = 0; pa <= kLegs; ++pa) {
                    M += part_at(pa, 0);
                    UNROLL_RBD for (int c = 0; c < 3; ++c) h1[c] += part_at(pa, 1 + c);
                    UNROLL_RBD for (int c = 0; c < 6; ++c) hO[c] += part_at(pa, 4 + c);
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
                UNROLL_RBD for (int f = 0; f < kFrameSlots; ++f) c += tk.frame_w(f) * 0.5 * q.ft[f];      // (q.ft: written above by this wave -- LDS keeps a wave's order)
                q.bc[m3s][0] = c;
            }
#ifdef BWD_PROFILE
            if (do_sum) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); fsum += __builtin_readcyclecounter() - fsum0; }
#endif
            if (run && do_chain && !terminal) {      // the next node's task block, requested at the top of this node
                UNROLL_RBD for (int k = 0; k < 3; ++k) { const int i = l + kFwdLanes * k; if (i <= kNodeTaskDoubles) q.tk[m3n][i] = ntk[k]; }
            }
#ifdef BWD_PROFILE
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); fwork += __builtin_readcyclecounter() - fnode0; }
#endif
            fwd_sync<NW>();
#ifdef BWD_PROFILE
            fwait += __builtin_readcyclecounter() - fnode0;
#endif
            if (run) {
                if (ts >= 0) {      // node ts's cost: residual terms + (state + control)
                    double c = q.bc[m3s][0] + (q.bc[m3s][2] + q.bc[m3s][3]);
                    if (ts != T) c *= kSplit ? dt_behind : dtn;
                    if (q.bc[m3s][1] != 0.0 || !(fabs(c) < INFINITY)) run = false;      // tryStep threw: this step length is out
                    ctry += c;
                }
                dt_behind = dtn;
                if (!terminal && do_chain) {     // (the chain wave's own: nobody else reads the nominal state)
                    UNROLL_RBD for (int k = 0; k < 3; ++k) { const int i = l + kFwdLanes * k; if (i < kNX) q.xs[i] = nxs[k]; }
                    if (per_node) stage_reg(t + 1);
                }
            }
            // ONE barrier per node: what the next node reads -- its state (written by the Euler step), its task block -- went into the
            // other slots before the barrier above, and what this node's end reads (the cost terms) sits in slots nobody writes
            // during the next node.  Only the per-node regularisation reference has a single copy.
            if (per_node) fwd_sync<NW>();
            FSTAMP(4)
        }
        bool pass = false;               // the trial ran to the end and passes SolverDDP's acceptance test
        if (live && run) {
            const double dV = cost - ctry;
            const double dVexp = al * (d1 + 0.5 * al * d2);
            pass = dVexp >= 0.0 && (d1 < 1e-12 || !feas || dV > 0.1 * dVexp);
        }
        if (!spec) {
            if (pass) { accepted = true; live = false; cost_try = ctry; }
        } else if (all10) {
            // this workgroup's results to the workspace; the last of the problem's three workgroups to arrive decides
            if (l == 0 && ia < 10) { ws[L.votes + 2 * ia] = pass ? 1.0 : 0.0; ws[L.votes + 2 * ia + 1] = ctry; }
            __threadfence();
            fwd_sync<NW>();
            unsigned *arrive = reinterpret_cast<unsigned *>(ws + L.arrive);
            if (threadIdx.x == 0) s.vote[0] = atomicAdd(arrive, 1u) == 2u ? 1.0 : 0.0;
            fwd_sync<NW>();
            if (s.vote[0] == 0.0) return;
            __threadfence();
            if (threadIdx.x == 0) *arrive = 0u;
            int w = -1;
            double vv[10];      // (the ten votes requested together: read in the loop below, each was waited for before the next was asked for)
            UNROLL_RBD for (int k = 0; k < 10; ++k) vv[k] = __builtin_nontemporal_load(ws + L.votes + 2 * k);
            UNROLL_RBD for (int k = 9; k >= 0; --k) if (vv[k] != 0.0) w = k;    // first in SolverDDP's order
            if (live) {
                if (w >= 0) { accepted = true; win = w; alpha = ldexp(1.0, -w); cost_try = __builtin_nontemporal_load(ws + L.votes + 2 * w + 1); }
                else alpha = ldexp(1.0, -9);
                live = false;
            }
            break;
        } else {
            if (do_chain && l == 0) { s.vote[si] = pass ? 1.0 : 0.0; s.ctry[si] = ctry; }      // (every wave holds the same numbers; one writes)
            fwd_sync<NW>();
            int w = -1;
            UNROLL_RBD for (int k = kFwdSub - 1; k >= 0; --k) if (s.vote[k] != 0.0) w = k;    // first in SolverDDP's order
            if (live && w < 0 && round == 0) widen = true;     // past the first four step lengths: flagged from now on
            if (live) {
                if (w >= 0) { accepted = true; win = w; alpha = ldexp(1.0, -(kFwdSub * round + w)); cost_try = s.ctry[w]; live = false; }
                else if (kFwdSub * (round + 1) >= 10) { alpha = ldexp(1.0, -9); live = false; }   // every step length tried
            }
            fwd_sync<NW>();
        }
    }
#ifdef BWD_PROFILE
    if (pvalid && lane == 0) { ws[L.Qu + 8 + 2 * wave] = (double)fwork; ws[L.Qu + 9 + 2 * wave] = (double)fwait; }   // tools/bwd_profile.py
    if (pvalid && lane == 0 && wave == 0) { for (int k = 0; k < 5; ++k) ws[L.Qu + 16 + k] = (double)fph[k]; }
    if (pvalid && lane == 0 && do_sum && wave != 0) ws[L.Qu + 22] = (double)fsum;      // the sum of the parts, on its wave
#endif
    if (!FUSED && (!owner || !do_chain)) return;
    if (owner && do_chain) {
    double xreg = sc[S_XREG];
    if (accepted) {   // setCandidate(xs_try, us_try, true)
        const long xsrc = L.xs_try + (long)win * nn * kNX, usrc = L.us_try + (long)win * T * kNV;
        // (eight elements per lane in flight: one by one, each load was waited for before its store -- 38 round trips to the
        // workspace per accepted step of the four-problems-per-wave mapping)
        const int i0 = spec ? lane : l, step = spec ? 64 : kFwdLanes;
        auto copy8 = [&](long dst, long src, long src0, long n0, long n) {      // ws[dst + i] = ws[(i < n0 ? src0 : src) + i], i = i0, i0 + step, ... < n
            for (long i = i0; i < n; i += 8L * step) {
                double v[8];
                UNROLL_RBD for (int k = 0; k < 8; ++k) { long j = i + (long)k * step; j = j < n ? j : n - 1; v[k] = ws[(j < n0 ? src0 : src) + j]; }      // (unconditional: a lane past the end reads the last element)
                UNROLL_RBD for (int k = 0; k < 8; ++k) { const long j = i + (long)k * step; if (j < n) ws[dst + j] = v[k]; }
            }
        };
        copy8(L.xs, xsrc, L.xs_try, kNX, (long)nn * kNX);
        copy8(L.us, usrc, usrc, 0, (long)T * kNV);
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
    __builtin_amdgcn_wave_barrier();
    if (spec ? lane == 0 : l == 0) {
        if (accepted) { sc[S_WASFEAS] = feas ? 1.0 : 0.0; sc[S_FEAS] = 1.0; sc[S_COST] = cost_try; sc[S_RECALC] = 1.0; }
        else sc[S_RECALC] = 0.0;
        sc[S_XREG] = xreg; sc[S_ITERS] = iters;
        if (!done && a.list && !FUSED) {      // goes on: onto the next iteration's list (the order there is arbitrary; nothing depends on it)
            const int nxt = (a.iter + 1) & 1;
            const int pos = atomicAdd(a.count + nxt, 1);
            if ((unsigned)pos < (unsigned)a.B) a.list[(long)nxt * a.B + pos] = (int)b;
            else index_error(a, IK_ERR_LIST_APPEND, pos);
            if (widen) sc[S_WIDE] = 1.0;
            double now = 0.0;
            if (a.wide && (widen || sc[S_WIDE] != 0.0)) {     // a place on the wide list while there are any
                const int at = atomicAdd(a.wcount + nxt, 1);
                if (at < kWideMax) { a.wide[nxt * kWideMax + at] = (int)b; now = 1.0; }
            }
            sc[S_WIDENOW] = now;
            if (a.near && sc[S_STOP] < a.near_stop) atomicAdd(a.near + nxt, 1);       // express-lane trigger statistics (ik_select_kernel)
        } else if (FUSED && !done && widen) sc[S_WIDE] = 1.0;
        if (iters <= (double)kTraceIters) {
            double *tr = ws + L.trace + ((long)iters - 1) * kTraceDoubles;
            tr[0] = accepted ? cost_try : cost; tr[1] = xreg; tr[2] = accepted ? alpha : 0.0; tr[3] = sc[S_STOP];
        }
        if (done) { sc[S_DONE] = 1.0; sc[S_STATUS] = status; atomicSub(a.active, 1); }
    }
    }
}

// The line-search kernels.  Each wave's role is a separate INSTANTIATION of forward_body (ROLES a template constant), all of
// them inline in the kernel behind a branch on the wave number.  History: with the roles as run-time conditions inside one
// instantiation hipcc's allocator let their register arrays interfere (512 registers + 124 spilled); as non-inlined functions per
// role that was gone, but a called function saves and restores the callee-saved registers it uses -- 314 scratch stores at entry
// and 314 loads before the return for the chain role, ~18 us of every launch (Go2 forward pass 25.5 -> 23.7 ms with the chain
// role inline); separate instantiations need no call at all (23.35 ms, no spills, 228-230 accumulation registers).
__shared__ ForwardLds g_fwd;
template <int NW>
__global__ __launch_bounds__(64 * NW) void ik_forward_kernel(const IkBatchArgs a) {
    if (NW == 1) {      // one wave does everything: nothing to separate
        forward_body<1, false, 15>(a, g_fwd, -1);
        return;
    }
    const int wave = threadIdx.x >> 6;
    if (wave == 0) forward_body<NW, false, forward_roles(NW, 0)>(a, g_fwd, -1);
    else if (wave == 1) forward_body<NW, false, forward_roles(NW, 1)>(a, g_fwd, -1);
    else forward_body<NW, false, forward_roles(NW, 2)>(a, g_fwd, -1);
}

// ------------------------------------------------------------------------------ fused ---
// One persistent workgroup of four waves (one per SIMD of a CU) runs whole DDP iterations of ONE problem without leaving the chip
// and without the host: derivative pass and Riccati pass pipelined over a common sequence of barriers (FusedCtl above), then the
// line search.  Same device code as the multi-kernel path (state_node, calc_*, backward_*_wave, forward_body), so a problem's
// result does not depend on which path -- or which mixture of the two -- solved it.  Used for (a) the EXPRESS LANE: the problems
// ik_select_kernel picks as stragglers-to-be leave the batch early and iterate here at their own pace, on a side stream, while
// the batch goes on in lock-step without them; (b) the tail: once few problems are left, all of them.
struct alignas(16) FusedLds {
    BackwardLds<2> bw;
    ForwardLds fw;              // fw.m: the robot model, staged once (the producers read it too)
    CalcNode nd[2][kCalcNodes]; // the two nodes each producer wave has in hand
    FusedCtl ctl;
    IkBatchArgs args;           // the launch arguments, for the role functions (below)
};
// File scope, so that the role functions below can name it (an LDS object reached through a generic reference would be read with
// flat instructions).  Only ik_fused_kernel uses it.
__shared__ FusedLds g_fused;

// The four roles are separate NON-INLINED functions: in one function body hipcc's register allocator lets the roles' big register
// arrays interfere (the Riccati wave's matrix rows, the producers' robot walk, the line search): 301 spilled registers for code
// that spills none (Riccati pass) to 124 (line search) as kernels of their own, and every phase ran 35-65 % slower than its
// kernel.  As functions each role has its own allocation.  They take no pointers: the launch arguments come out of LDS, every
// field made wave-uniform again (v_readfirstlane: scalar registers, scalar branches, as kernel arguments are).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uni(long v) {
    return (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)v >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)v));
}
template <class P> __device__ __forceinline__ P *uni(P *p) { return reinterpret_cast<P *>(uni((long)p)); }
__device__ __forceinline__ IkBatchArgs uniform_args(const IkBatchArgs &g) {
    IkBatchArgs a;
    a.B = uni(g.B); a.T = uni(g.T); a.maxiter = uni(g.maxiter); a.fwd_spec = uni(g.fwd_spec); a.bwd_waves = uni(g.bwd_waves);
    a.list = uni(g.list); a.count = uni(g.count); a.wide = uni(g.wide); a.wcount = uni(g.wcount); a.err = uni(g.err); a.near = uni(g.near);
    a.xmeta = uni(g.xmeta); a.xlist = uni(g.xlist); a.iter = uni(g.iter); a.n_launch = uni(g.n_launch);
    a.model = uni(g.model); a.x0 = uni(g.x0); a.dt = uni(g.dt); a.tasks = uni(g.tasks); a.state_w = uni(g.state_w); a.x_reg = uni(g.x_reg);
    a.ctrl_w = uni(g.ctrl_w); a.s_state_w = uni(g.s_state_w); a.s_ctrl_w = uni(g.s_ctrl_w); a.s_x_reg = uni(g.s_x_reg);
    a.sn_state_w = uni(g.sn_state_w); a.sn_x_reg = uni(g.sn_x_reg); a.sn_ctrl_w = uni(g.sn_ctrl_w); a.ws = uni(g.ws); a.active = uni(g.active); a.near_stop = g.near_stop;
    return a;
}
__device__ __forceinline__ IkBatchArgs fused_args() { return uniform_args(g_fused.args); }

// Producer wave p (0 / 1) of the fused kernel: node pairs p, p + 2, ... counted from the terminal node down (pair j = nodes
// T - 2j, T - 2j - 1), three ticks per pair -- [stage, walk] [columns, state terms in, assemble A] [assemble B] -- each shorter than
// a node of the recursion, which then never waits for a producer at a tick (with walk and columns in one tick of 23K cycles it
// did, 5K cycles per node: measured with idle producers); together the producers deliver four nodes per three ticks against the
// recursion's one per tick.  Producer 1 spends its first tick on the scalar chains of every node (one lane per node).  The
// walk's per-lane state reaches the columns in REGISTERS, as in ik_calcdiff_kernel (parked in LDS in between, the compiler can no
// longer contract a product of the walk into a sum of the columns and the derivatives differ in their last bits from the
// multi-kernel path's), and the pair's three steps are STRAIGHT-LINE code around their ticks: as states of a loop around one
// barrier, hipcc carried that state across the back edge in scratch.
__device__ __forceinline__ void producer_wave(const IkBatchArgs &a, long b, FusedLds &s, int p, int lane, Ticker &tk) {
    FusedCtl &ctl = s.ctl;
    const int stamp = ctl.stamp, T = a.T, nn = T + 1, npairs = (nn + 1) / 2;
    const IkLayout L = IkLayout::make(T);
    double *ws = a.ws + b * L.total;
#ifdef FUSED_NO_PRODUCE     // timing experiment only (wrong results): the producers do nothing, the derivatives of the lock-step iterations stay
    const bool recalc = false;
#else
    const bool recalc = ws[L.scal + S_RECALC] != 0.0;
#endif
    const RobotModelDev &m = s.fw.m;
    const double *state_w0 = batch_ptr(a.state_w, a.s_state_w, b), *ctrl_w0 = batch_ptr(a.ctrl_w, a.s_ctrl_w, b);
    const int hs = lane >> 5, hl = lane & 31;
    CalcNode &qw = s.nd[p][hs];
    // A flag is raised only once the global stores it stands for have completed: the recursion reads flags in the middle of ticks
    // too (whether node t - 1 can be requested a node ahead), not only right behind the barrier that follows the producer's step.
    auto stores_done = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // one tick; true when the phase is over (or the watchdog fired): the caller returns at once, with no further barrier
    auto tick_over = [&]() -> bool { tick_mem(tk); return tk.dead || tk.n >= lds_flag(ctl.done_tick); };
    if (!recalc) {
        // the line search of the iteration before accepted nothing: the derivatives in the workspace are still those of this trajectory
        if (p == 0) { if (lane <= T) ctl.ready[lane] = stamp; if (lane == 0) ctl.state_ready = stamp; }
    } else {
        if (p == 1) {
            if (lane <= T) state_node<0>(a, b, lane, ws, L);
            stores_done();
            if (lane == 0) ctl.state_ready = stamp;
            if (tick_over()) return;
        }
        for (int pair = p; pair < npairs; pair += 2) {
            const int tA = T - 2 * pair, tB = tA - 1, tw = hs == 0 ? tA : tB;
            const bool wvalid = tw >= 0;
            UNROLL_RBD for (int h = 0; h < kCalcNodes; ++h) { const int t = h == 0 ? tA : tB; if (t >= 0) calc_stage_simple(a, b, t, ws, L, s.nd[p][h], lane); }
            wave_sync();
            NodeTasks tkw{qw.tk};
            PartWalk pw;
            double Rb[9], pb[3], Vb[6];
            if (wvalid && hl < kNV) calc_walk(m, qw, tkw, hl, Rb, pb, Vb, pw);
            if (tick_over()) return;
            // (the scalar chains of these nodes are there: they take producer 1's first tick, and this is tick 2 at the earliest)
            if (wvalid) calc_fetch_state(qw, ws, L, tw, hl);
            if (wvalid && hl < kNV) calc_columns(m, qw, tkw, hl, pw);
            wave_sync();
            calc_assemble(a, b, tA, s.nd[p][0], ws, L, lane, state_w0, ctrl_w0, &ctl.node_cost[tA]);
            stores_done();
            if (lane == 0) ctl.ready[tA] = stamp;
            if (tick_over()) return;
            if (tB >= 0) {
                calc_assemble(a, b, tB, s.nd[p][1], ws, L, lane, state_w0, ctrl_w0, &ctl.node_cost[tB]);
                stores_done();
                if (lane == 0) ctl.ready[tB] = stamp;
                if (tick_over()) return;
            }
        }
    }
    for (;;) { if (tick_over()) return; }      // everything produced: ticks until the phase is over
}

// the roles: each returns the ticks it counted, or -1 when the watchdog fired (every wave counts the same barriers)
__device__ __noinline__ int fused_role_recursion(long b_, int limit) {
    const IkBatchArgs a = fused_args();
    Ticker tk; tk.limit = uni(limit);
    backward_main_wave<2, true>(a, uni(b_), g_fused.bw, (int)(threadIdx.x & 63), &g_fused.ctl, tk);
    if ((threadIdx.x & 63) == 0) g_fused.ctl.wait[0] += tk.wait;
    return tk.dead ? -1 : tk.n;
}
__device__ __noinline__ int fused_role_gains(long b_, int limit) {
    const IkBatchArgs a = fused_args();
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + uni(b_) * L.total;
    Ticker tk; tk.limit = uni(limit);
    backward_gains_wave<true>(g_fused.bw, ws, ws + L.scal, L, a.T, (int)(threadIdx.x & 63), &g_fused.ctl, tk);
    if ((threadIdx.x & 63) == 0) g_fused.ctl.wait[1] += tk.wait;
    return tk.dead ? -1 : tk.n;
}
__device__ __noinline__ int fused_role_producer(long b_, int limit) {
    const IkBatchArgs a = fused_args();
    Ticker tk; tk.limit = uni(limit);
    producer_wave(a, uni(b_), g_fused, uni((int)(threadIdx.x >> 6)) - 2, (int)(threadIdx.x & 63), tk);
    if ((threadIdx.x & 63) == 0) g_fused.ctl.wait[threadIdx.x >> 6] += tk.wait;
    return tk.dead ? -1 : tk.n;
}
template <int ROLES>
__device__ __noinline__ void fused_role_line_search(long b_) {
    const long long t0 = __builtin_readcyclecounter();
    const IkBatchArgs a = fused_args();
    forward_body<4, true, ROLES>(a, g_fused.fw, uni(b_));
    if (threadIdx.x == 0) g_fused.ctl.tele[3] += (long long)__builtin_readcyclecounter() - t0;
}

// plist[0 .. *pcount) (at most gridDim.x of them): the problems to run to the end.  gate: when given, the launch is void
// unless gate[0] == 1 and gate[2] == gate_iter (the express lane: ik_select_kernel decides on the device whether, and when,
// it starts; the host enqueues the pair after every early iteration).
__global__ __launch_bounds__(256) void ik_fused_kernel(const IkBatchArgs a, const int *plist, const int *pcount, const int *gate, int gate_iter) {
    FusedLds &s = g_fused;
    if (gate && (gate[0] != 1 || gate[2] != gate_iter)) return;
    const int n = *pcount;
    if ((int)blockIdx.x >= n) return;
    const int bi = plist[blockIdx.x];
    if ((unsigned)bi >= (unsigned)a.B) { if (threadIdx.x == 0) index_error(a, IK_ERR_LIST_ENTRY, bi); return; }
    const long b = bi;
    const int wave = threadIdx.x >> 6;
    const IkLayout L = IkLayout::make(a.T);
    double *ws = a.ws + b * L.total;
    double *sc = ws + L.scal;
    {
        stage_model<256>(a.model, &s.fw.m);
    }
    if (threadIdx.x < 64) s.ctl.ready[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s.ctl.state_ready = 0; s.ctl.hand_count = 0; s.ctl.done_tick = 0x7fffffff; s.ctl.stamp = 0; s.args = a;
                            for (int k = 0; k < 4; ++k) { s.ctl.tele[k] = 0; s.ctl.wait[k] = 0; } }
    __syncthreads();
    const int limit = 24 * (a.T + 8);      // a pass takes T + ~6 ticks; the regularisation can restart it 18 times (1e-9 ... 1e9)
    long long cyc_a = 0, cyc_b = 0, ticks = 0, turns = 0;       // telemetry (tools/ik_run.py): cycles in the two phases, ticks, iterations
    for (int turn = 0; turn <= a.maxiter; ++turn) {      // one DDP iteration per turn (the forward pass ends the problem at maxiter)
        if (sc[S_DONE] != 0.0) break;
        const long long c0 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) { s.ctl.stamp += 1; s.ctl.hand_count = 0; s.ctl.done_tick = 0x7fffffff; s.ctl.t_phase = c0; }
        __syncthreads();
        const int nt = wave == 0 ? fused_role_recursion(b, limit) : wave == 1 ? fused_role_gains(b, limit) : fused_role_producer(b, limit);
        if (nt < 0) {      // every wave has counted the same barriers: all of them are here, none waits at one
            if (threadIdx.x == 0) index_error(a, IK_ERR_FUSED_WATCHDOG, (int)b);
            return;
        }
        __syncthreads();
        const long long c1 = __builtin_readcyclecounter();
        cyc_a += c1 - c0; ticks += nt; ++turns;
        if (sc[S_DONE] != 0.0) break;       // the pass was given up at the regularisation's ceiling
        if (wave == 0) {       // the chain role inline, as in the line-search kernels (no callee-saved registers to save and restore)
            forward_body<4, true, forward_roles(4, 0)>(a, s.fw, b);
            if (threadIdx.x == 0) s.ctl.tele[3] += (long long)__builtin_readcyclecounter() - c1;
        } else if (wave == 1) fused_role_line_search<forward_roles(4, 1)>(b);
        else if (wave == 2) fused_role_line_search<forward_roles(4, 2)>(b);
        else fused_role_line_search<forward_roles(4, 3)>(b);
        __syncthreads();
        cyc_b += __builtin_readcyclecounter() - c1;
    }
    if (threadIdx.x == 0) {     // (the Qu slot of the workspace is unused by the solver)
        ws[L.Qu + 0] = (double)cyc_a; ws[L.Qu + 1] = (double)cyc_b; ws[L.Qu + 2] = (double)ticks; ws[L.Qu + 3] = (double)turns;
        for (int k = 0; k < 4; ++k) { ws[L.Qu + 4 + k] = (double)s.ctl.tele[k]; ws[L.Qu + 8 + k] = (double)s.ctl.wait[k]; }
    }
}

// The express lane's selection, one workgroup, in front of iteration a.iter.  It acts ONCE per batch solve, at the first call
// that finds the batch where the lane pays: (almost) nobody has finished yet, at least half of the problems are within reach of
// the stopping threshold (|Q_u|^2 < near_stop at their last Riccati pass) and an iteration earlier (almost) none was -- the batch
// crossed that line in ONE iteration: it converges fast, it will be gone in two or three iterations, and what is far from
// converging now will still be iterating long after.  (Measured on the CPU twin's traces: in the Solo12 H_ik = 10 batch the median
// |Q_u|^2 falls 21 -> 0.07 -> 1e-4 over iterations 2, 3, 4, and after iteration 3 the 24 longest-running problems, 12 ... 36
// iterations, are all among the 110 largest |Q_u|^2; the synthetic Go2 H_ik = 30 batch converges slowly everywhere, the share of
// problems below the line creeps up over ten iterations, |Q_u|^2 does not rank its long runs -- and the lane stays shut.)  The
// `cap` problems with the largest |Q_u|^2 move from the active list to xlist; the fused kernel enqueued behind this one on the
// side stream takes them.  xmeta = {state (1: taken), count, iteration, -}.
__global__ __launch_bounds__(1024) void ik_select_kernel(const IkBatchArgs a, int cap, int force) {
    __shared__ unsigned hist[16];
    __shared__ unsigned long long prefix_s;
    __shared__ int want_s, nx_s, nk_s, go_s;
    const int cur = a.iter & 1, tid = threadIdx.x;
    const IkLayout L = IkLayout::make(a.T);
    if (tid == 0) {
        const int n = a.count[cur];
        int c = cap < kExpressMax ? cap : kExpressMax;
        if (c > n / 8) c = n / 8;          // never more than an eighth of what is left
        go_s = a.xmeta[0] == 0 && c > 0 && (unsigned)n <= (unsigned)a.B &&
               (force || ((long)n * 50 >= (long)a.B * 49 && (long)a.near[cur] * 2 >= (long)a.B &&
                          (long)a.near[cur ^ 1] * 20 <= (long)a.B));     // (near[cur ^ 1]: the iteration before; this iteration's state kernel resets it)
        want_s = c; prefix_s = 0ull; nx_s = 0; nk_s = 0;
    }
    __syncthreads();
    if (!go_s) return;
    const int n = a.count[cur], want = want_s;
    const int *lst = a.list + (long)cur * a.B;
    int *other = a.list + (long)(cur ^ 1) * a.B;       // free until this iteration's forward pass appends to it
    auto key_of = [&](int i) -> unsigned long long {   // |Q_u|^2 >= 0: the bit pattern orders like the value (NaN above everything)
        const int p = lst[i];
        const double v = (unsigned)p < (unsigned)a.B ? a.ws[(long)p * L.total + L.scal + S_STOP] : 0.0;
        return (unsigned long long)__double_as_longlong(v < 0.0 ? 0.0 : v);
    };
    // radix select, four bits at a time from the top: prefix = the want-th largest key
    int remaining = want;
    for (int shift = 60; shift >= 0; shift -= 4) {
        if (tid < 16) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long pre = prefix_s, mask = shift == 60 ? 0ull : (~0ull << (shift + 4));
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long k = key_of(i);
            if ((k & mask) == pre) atomicAdd(&hist[(k >> shift) & 15u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int d = 15, acc = 0;
            for (; d > 0; --d) { if (acc + (int)hist[d] >= remaining) break; acc += (int)hist[d]; }
            remaining -= acc;
            prefix_s = pre | ((unsigned long long)d << shift);
            want_s = remaining;      // of the keys equal to the final prefix, this many are taken
        }
        __syncthreads();
        remaining = want_s;
    }
    const unsigned long long kth = prefix_s;
    // above the threshold: taken; equal to it: the first `remaining` to arrive; the rest stays on the list
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long k = key_of(i);
        const int p = lst[i];
        bool take = k > kth;
        if (!take && k == kth) take = atomicAdd(&nk_s, 1) < remaining;
        if (take) { const int at = atomicAdd(&nx_s, 1); if (at < kExpressMax) a.xlist[at] = p; else take = false; }
        if (take && a.wide) {
            // a taken problem that holds a place of the wide list gives it up (-1 = an empty place): the two extra workgroups of this
            // iteration's line search would otherwise read its trajectories and gains while the fused kernel rewrites them on
            // the side stream.  (S_WIDENOW stays: with it cleared an extra workgroup would behave as a regular one.)
            const int nw = a.wcount[cur] < kWideMax ? a.wcount[cur] : kWideMax;
            for (int e = 0; e < nw; ++e) if (a.wide[cur * kWideMax + e] == p) a.wide[cur * kWideMax + e] = -1;
        }
        if (!take) other[atomicAdd(&go_s, 1) - 1] = p;         // go_s was 1: it now counts the kept ones + 1
    }
    __syncthreads();
    const int kept = go_s - 1;
    for (int i = tid; i < kept; i += 1024) const_cast<int *>(lst)[i] = other[i];
    if (tid == 0) {
        a.count[cur] = kept;
        a.xmeta[1] = nx_s < kExpressMax ? nx_s : kExpressMax; a.xmeta[2] = a.iter;
        __threadfence();
        a.xmeta[0] = 1;
    }
}

// ------------------------------------------------------------ small helper kernels ---
// (64-thread workgroups with the whole register file: at the default bound hipcc built these two for 128 registers, 570 values in scratch)
__global__ __launch_bounds__(64) void ik_centroidal_state_kernel(const RobotModelDev *model, const double *x, double *out9, int B) {
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

__global__ __launch_bounds__(64) void ik_com_mom_kernel(const RobotModelDev *model, const double *xs, double *com, double *mom, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pass1 p1;
    const int fid[kFrameSlots] = {-1, -1, -1, -1};
    quad_pass1<false>(*model, xs + i * kNX, fid, p1);
    for (int c = 0; c < 3; ++c) com[i * 3 + c] = p1.com[c];
    for (int c = 0; c < 6; ++c) mom[i * 6 + c] = p1.hg[c];
}

// both versions of the state operators on n samples (tests/test_rbd_gpu.py)
__global__ void ik_state_ops_selftest_kernel(const double *x0, const double *x1, const double *dx, int n, double *dq, double *dr, double *iq, double *ir) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[kNX], b[kNX], d[kNDX], o[kNX];
    for (int k = 0; k < kNX; ++k) { a[k] = x0[(long)i * kNX + k]; b[k] = x1[(long)i * kNX + k]; }
    for (int k = 0; k < kNDX; ++k) d[k] = dx[(long)i * kNDX + k];
    double r[kNDX];
    state_diff_q(a, b, r);
    for (int k = 0; k < kNDX; ++k) dq[(long)i * kNDX + k] = r[k];
    state_diff<false>(a, b, r, nullptr);
    for (int k = 0; k < kNDX; ++k) dr[(long)i * kNDX + k] = r[k];
    state_integrate_q(a, d, o);
    for (int k = 0; k < kNX; ++k) iq[(long)i * kNX + k] = o[k];
    state_integrate(a, d, o);
    for (int k = 0; k < kNX; ++k) ir[(long)i * kNX + k] = o[k];
}

// the active-problem counter, copied to a host-mapped word: the host reads it after its stream synchronisation
// without a device-to-host copy operation (a 4-byte hipMemcpy into pageable memory costs ~50 us per DDP iteration)
__global__ void ik_publish_active_kernel(const int *active, const int *err, const int *count_next, const int *xmeta, volatile int *host_word,
                                         const double *ws, long ws_stride, long scal, int n_iters_of) {
    if (n_iters_of > 0) {      // the fused-direct path (a handful of problems): the most DDP iterations any of them ran
        int mx = 0;
        for (int b = 0; b < n_iters_of; ++b) { const int n = (int)ws[b * ws_stride + scal + S_ITERS]; mx = n > mx ? n : mx; }
        host_word[4] = mx;
    }
    host_word[1] = err ? err[0] : 0;
    host_word[2] = count_next ? *count_next : -1;       // problems on the list the next iteration runs over (the express lane's are not)
    host_word[3] = xmeta ? xmeta[0] : 0;                // the express lane has taken its problems
    host_word[0] = *active;
    __threadfence_system();
}

// com / momentum references of the IK tracking tasks from the centroidal solution X
// (KinoDynMP::optimize, kino_dyn.cpp:50-56: rows 0..T-1 running, row T terminal; mom = [m v, L])
__global__ void kd_fill_refs_kernel(double *tasks, const double *X, double m, int B, int H, int T) {
    // one thread per ELEMENT (nine per node: neighbouring lanes write neighbouring words; one thread per node wrote nine words at a
    // stride of 264 bytes each: 190 us for 45 K nodes)
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 9L * B * (T + 1)) return;
    const long node = id / 9;
    const int e = (int)(id % 9), g = e / 3, c = e % 3;      // g: 0 CoM, 1 linear momentum (m v), 2 angular momentum
    const long b = node / (T + 1);
    const int t = (int)(node % (T + 1));
    const double v = X[b * 9L * (H + 1) + 9L * t + e];
    tasks[node * kNodeTaskDoubles + 5 * kFrameSlots + (g == 0 ? 1 : g == 1 ? 5 : 8) + c] = g == 1 ? m * v : v;
}

}  // namespace

hipError_t ik_launch_fill_refs(double *tasks, const double *X, double m, int B, int H, int T, hipStream_t st) {
    const long n = 9L * B * (T + 1);
    hipLaunchKernelGGL(kd_fill_refs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tasks, X, m, B, H, T);
    return hipGetLastError();
}
hipError_t ik_launch_state_ops_selftest(const double *x0, const double *x1, const double *dx, int n, double *dq, double *dr, double *iq, double *ir,
                                        hipStream_t st) {
    hipLaunchKernelGGL(ik_state_ops_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, st, x0, x1, dx, n, dq, dr, iq, ir);
    return hipGetLastError();
}
hipError_t ik_launch_publish_active(const IkBatchArgs &a, int next_iter, int *host_word_dev, hipStream_t st, int n_iters_of) {
    const IkLayout L = IkLayout::make(a.T);
    hipLaunchKernelGGL(ik_publish_active_kernel, dim3(1), dim3(1), 0, st, a.active, a.err, a.count ? a.count + (next_iter & 1) : nullptr, a.xmeta, host_word_dev,
                       a.ws, L.total, L.scal, n_iters_of);
    return hipGetLastError();
}
hipError_t ik_launch_init(const IkBatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ik_init_kernel, dim3((a.B + 63) / 64), dim3(64), 0, st, a);
    return hipGetLastError();
}
// launches cover the problems of the active list (a.n_launch, the host's last look at the counter, bounds its length)
static long launch_problems(const IkBatchArgs &a) { return a.list ? (a.n_launch < a.B ? a.n_launch : a.B) : a.B; }
hipError_t ik_launch_state(const IkBatchArgs &a, hipStream_t st) {
    const long n = launch_problems(a) * (a.T + 1);
    hipLaunchKernelGGL(ik_state_kernel, dim3(2u * (unsigned)((n + 63) / 64)), dim3(64), 0, st, a);      // two workgroups per 64 nodes
    return hipGetLastError();
}
int g_calcdiff_one_wave_above = 1024;     // node pairs per launch above which one wave takes a pair (the two-wave kernel has 1024 pairs resident on an MI355X)
hipError_t ik_launch_calcdiff(const IkBatchArgs &a, hipStream_t st) {
    const long n = launch_problems(a) * ((a.T + 1 + 1) / 2);   // two nodes per workgroup
    if (n > g_calcdiff_one_wave_above) hipLaunchKernelGGL(ik_calcdiff1_kernel, dim3((unsigned)((n + 1) / 2)), dim3(128), 0, st, a);
    else hipLaunchKernelGGL(ik_calcdiff_kernel, dim3((unsigned)n), dim3(128), 0, st, a);
    return hipGetLastError();
}
// workgroups of each IK kernel one CU holds at once, as the runtime's occupancy query sees them (registers, LDS, wave slots):
// {calcdiff (two waves per pair), calcdiff1 (one wave per pair), backward<1>, backward<2>, forward<1>, forward<2>, forward<3>, state}
void ik_kernel_occupancy(int *out8) {
    for (int i = 0; i < 8; ++i) out8[i] = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[0], ik_calcdiff_kernel, 128, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[1], ik_calcdiff1_kernel, 128, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[2], ik_backward_kernel<1>, 64, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[3], ik_backward_kernel<2>, 128, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[4], ik_forward_kernel<1>, 64, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[5], ik_forward_kernel<2>, 128, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[6], ik_forward_kernel<3>, 192, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out8[7], ik_state_kernel, 64, 0);
}
hipError_t ik_launch_backward(const IkBatchArgs &a, hipStream_t st) {
    if (a.bwd_waves == 2) hipLaunchKernelGGL(ik_backward_kernel<2>, dim3((unsigned)launch_problems(a)), dim3(128), 0, st, a);
    else hipLaunchKernelGGL(ik_backward_kernel<1>, dim3((unsigned)launch_problems(a)), dim3(64), 0, st, a);
    return hipGetLastError();
}
hipError_t ik_launch_forward(const IkBatchArgs &a, hipStream_t st) {
    const unsigned n = (unsigned)launch_problems(a);
    if (a.fwd_spec == 4) hipLaunchKernelGGL(ik_forward_kernel<3>, dim3(3 * n), dim3(192), 0, st, a);
    else if (a.fwd_spec == 3) hipLaunchKernelGGL(ik_forward_kernel<3>, dim3(n + (a.wide ? 2 * kWideMax : 0)), dim3(192), 0, st, a);
    else if (a.fwd_spec == 2) hipLaunchKernelGGL(ik_forward_kernel<2>, dim3(n + (a.wide ? 2 * kWideMax : 0)), dim3(128), 0, st, a);
    else if (a.fwd_spec == 1) hipLaunchKernelGGL(ik_forward_kernel<1>, dim3(n), dim3(64), 0, st, a);      // four step lengths of one problem per wave, every role on that wave
    else hipLaunchKernelGGL(ik_forward_kernel<1>, dim3((n + 3) / 4), dim3(64), 0, st, a);
    return hipGetLastError();
}
// the express lane (a.iter = the iteration about to start): selection on `st`, the fused kernel for what it took on `side`
hipError_t ik_launch_select(const IkBatchArgs &a, int cap, int force, hipStream_t st) {
    hipLaunchKernelGGL(ik_select_kernel, dim3(1), dim3(1024), 0, st, a, cap, force);
    return hipGetLastError();
}
hipError_t ik_launch_fused_express(const IkBatchArgs &a, int cap, hipStream_t side) {
    hipLaunchKernelGGL(ik_fused_kernel, dim3((unsigned)(cap < kExpressMax ? cap : kExpressMax)), dim3(256), 0, side, a, a.xlist, a.xmeta + 1, a.xmeta, a.iter);
    return hipGetLastError();
}
// the tail: every problem still on the active list of iteration a.iter (at most a.n_launch of them) to the end
hipError_t ik_launch_fused_tail(const IkBatchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ik_fused_kernel, dim3((unsigned)launch_problems(a)), dim3(256), 0, st, a, a.list + (long)(a.iter & 1) * a.B, a.count + (a.iter & 1),
                       static_cast<const int *>(nullptr), 0);
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

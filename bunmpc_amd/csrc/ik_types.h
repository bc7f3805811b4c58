// Internal types shared by the IK-DDP kernels (ik_ddp.hip) and the C-ABI host code.
// Restates, for the GPU, what the reference obtains from pinocchio 2.6.9 / crocoddyl 1.9.0 through
// ISL/src/ik/*.cpp (see oracle/rbd_np.py and oracle/ik_ddp_np.py for the CPU restatement).
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace bunmpc {

// records the thread-local text bmpc_last_error() returns; returns `code` (bunmpc_capi.hip)
int set_error(int code, const std::string &msg);

constexpr int kMaxJoints = 12;   // revolute joints after the free-flyer (serial chains off the base)
constexpr int kMaxFrames = 40;   // (Solo12: 35 frames, Go2: 39; the table is part of the model copy every derivative workgroup keeps in LDS)
constexpr int kNV = 18, kNQ = 19, kNX = 37, kNDX = 36;   // the kernels are built for 12 joints
constexpr int kFrameSlots = 4;   // frame-translation tasks per node (the harness adds <= n_eff)

// Rigid-body model (device copy).  Body 0 = base, body i+1 = joint i.
struct alignas(16) RobotModelDev {
    // one packed record per body for the kernels' batched reads (rbd_quad.h::load_body): [joint placement p (3) |
    // axis (3) | mass | com (3) | inertia xx xy xz yy yz zz (6)] = 16 doubles; body 0 (base) has no joint
    double rec[kMaxJoints + 1][16];
    int nj, nframes;
    int parent[kMaxJoints];          // -1 = base, else joint index
    int chain_end[kMaxJoints];       // last joint of the serial chain joint i belongs to
    int R_identity[kMaxJoints];      // joint placement rotation is the identity (URDF rpy = 0)
    double R[kMaxJoints][9], p[kMaxJoints][3], axis[kMaxJoints][3];
    double mass[kMaxJoints + 1], com[kMaxJoints + 1][3], inertia[kMaxJoints + 1][6];  // xx xy xz yy yz zz
    int frame_body[kMaxFrames];
    double frame_p[kMaxFrames][3];
    double total_mass;
};

// Per-problem task description, one block per node t = 0..T (T = terminal):
//   frame slots: weight, frame index (as double), ref(3)           -> 5 doubles x kFrameSlots
//   com:  weight, ref(3)                                            -> 4
//   mom:  weight, ref(6)                                            -> 7
//   state reg weight, ctrl reg weight                               -> 2
constexpr int kNodeTaskDoubles = 5 * kFrameSlots + 4 + 7 + 2;

// trial trajectories per problem: slot 0 is the one the sequential line search uses; the speculative line search
// (four step lengths of one problem at once, ik_forward_kernel) fills all four
constexpr int kTrySlots = 12;    // 10 used by the all-step-lengths mapping (three workgroups of four trials per problem)
// per-iteration trace of a problem's DDP (telemetry, read back by the parity tests): after iteration i (0-based, i < kTraceIters)
// [cost, regularisation, accepted step length (0 = none), stopping criterion] as SolverDDP holds them at the end of the iteration
constexpr int kTraceIters = 128, kTraceDoubles = 4;

// What the derivative pass hands the Riccati pass per node INSTEAD of the full Gauss-Newton L_xx (36 x 36 = 1296 doubles): the
// reference's cost set (ISL/src/ik/{com_tasks,end_effector_tasks,regularization_costs}.cpp, action_model.cpp:60-63,82-86) gives
//     L_xx = sc ( [L_qq' 0; 0 0]  +  wm M^T M  +  diag(0, wst sw_v) ),
// M = d h_g / d (q, v) (6 x 36: the only residual that sees the velocities), L_qq' (18 x 18) = CoM and frame-translation
// terms + the q-part of the state regularisation (its Jlog6 block on the free-flyer), sw_v the state weights of the velocities.
// L_xx enters the recursion only through V_xx = (L_xx + G) - Y^T Y, a sum taken on the matrix pipe in MFMA tiles: so L_qq'
// travels IN THAT TILE LAYOUT (kLqqDoubles: tile (0,0) as [v][lane] = 256, the two q-columns of tile (0,1) = 32, the 2 x 2
// corner of tile (1,1) = 4, padding) and M row-major with its weight and the velocity diagonal as the tiles need it
// (kHnDoubles: M [6][36] = 216 | sc wm | pad | d11 [16] at 224: diagonal entries of tile (1,1) by column | d22 [4] at 240: of
// tile (2,2)).  544 doubles per node instead of 1296, none of them read with a 288-byte stride.
constexpr int kLqqDoubles = 296, kHnDoubles = 248;
constexpr int kHnW = 216, kHnD11 = 224, kHnD22 = 240;

// Layout of the per-problem DDP workspace in HBM (doubles), T = number of running nodes.
struct IkLayout {
    int T;
    long xs, us, xs_try, us_try, fs, xnext, Lx, Lqq, Hn, Lu, Luu, A6, B6, K, kff, Qu, Quuk, scal, nrs, njl, ncs, trace, votes, arrive, total;
    __host__ __device__ static IkLayout make(int T) {
        IkLayout l; l.T = T;
        long o = 0;
        auto take = [&](long n) { long r = o; o += n; return r; };
        l.xs = take((long)(T + 1) * kNX); l.us = take((long)T * kNV);
        l.xs_try = take((long)kTrySlots * (T + 1) * kNX); l.us_try = take((long)kTrySlots * T * kNV);
        l.fs = take((long)(T + 1) * kNDX); l.xnext = take((long)T * kNX);
        l.Lx = take((long)(T + 1) * kNDX); l.Lqq = take((long)(T + 1) * kLqqDoubles); l.Hn = take((long)(T + 1) * kHnDoubles);
        l.Lu = take((long)T * kNV); l.Luu = take((long)T * kNV);
        l.A6 = take((long)T * 36); l.B6 = take((long)T * 36);
        l.K = take((long)T * kNV * kNDX); l.kff = take((long)T * kNV);
        l.Qu = take((long)T * kNV); l.Quuk = take((long)T * kNV);
        l.scal = take(16);
        // per node, written by ik_state_kernel for ik_calcdiff_kernel (bulk iterations): state residual, its Jlog6 block, cost part
        l.nrs = take((long)(T + 1) * kNDX); l.njl = take((long)(T + 1) * 36); l.ncs = take((long)T + 1);
        l.trace = take((long)kTraceIters * kTraceDoubles);
        // all-step-lengths line search: [pass flag, trial cost] per step length, and the arrival counter of the problem's workgroups
        l.votes = take(2 * kTrySlots); l.arrive = take(2);
        l.total = o;
        return l;
    }
};
// scalars kept per problem in ws[scal + i]
enum IkScal { S_COST = 0, S_XREG, S_D1, S_D2, S_STOP, S_FEAS, S_WASFEAS, S_DONE, S_ITERS, S_RECALC, S_STATUS, S_NODECOST,
              S_WIDE,       // sticky: a line search of this problem once needed more than the four step lengths of one workgroup
              S_WIDENOW };  // its next forward pass runs as three workgroups (it holds one of the kWideMax places of the wide list)
// Problems whose line search goes past four step lengths are few and always the same ones (Go2 H = 60, 1024 problems: three
// problems cause a second round in 76 of the 100 iterations), and a second round costs the whole batch a rollout's latency.
// Such a problem is flagged (S_WIDE) and from then on gets all ten step lengths at once, on three workgroups.
constexpr int kWideMax = 32;
// The express lane (ik_select_kernel / ik_fused_kernel in ik_ddp.hip): at most this many problems leave the batch early
constexpr int kExpressMax = 256;
// ints of device scratch behind bmpc_ik_batch_t.active_list: list[2][B], count[2], wide_count[2], wide[2][kWideMax], err[2],
// near[2], xmeta[4], xlist[kExpressMax]
inline long active_list_ints(long B) { return 2 * B + 4 + 2 * kWideMax + 2 + 2 + 4 + kExpressMax; }
// Index checks of the list code (always on: a few scalar compares per workgroup).  A list entry, a list length or an append
// position outside its range is never used as an index: the kernel records the code in err[0] (first one wins), drops the
// access, and the DDP loop returns BMPC_DEVICE_ERROR with it instead of the process dying in a memory fault.
enum IkIndexError { IK_ERR_NONE = 0, IK_ERR_LIST_ENTRY = 1, IK_ERR_LIST_COUNT = 2, IK_ERR_LIST_APPEND = 3, IK_ERR_WIDE_ENTRY = 4,
                    IK_ERR_FUSED_WATCHDOG = 5 };   // the fused kernel's tick watchdog (a protocol bug, never a data condition)

struct IkBatchArgs {
    int B, T, maxiter;
    int fwd_spec;              // forward pass: 0 = four problems per wave; 2 / 3 = one problem per workgroup of 2 / 3 waves,
                               // four step lengths at once (few active problems); 4 = three such workgroups per problem, all
                               // ten step lengths at once (very few active problems)
    // Active-problem list (or null: every launch covers all B problems, finished ones return at once).  Two ping-pong
    // lists of B problem indices + two counts: DDP iteration k works on list[k & 1][0 .. count[k & 1]) and its forward pass
    // appends the problems that go on to list[(k + 1) & 1].  n_launch (host side, one look behind) bounds count.
    int bwd_waves;             // backward pass: 1 = one wave per problem; 2 = a second wave for the gains (few active problems)
    int *list, *count;
    int *wide, *wcount;        // the flagged problems among them (see kWideMax): wide[k & 1][0 .. min(wcount[k & 1], kWideMax))
    int *err;                  // [2] index-check record of the list code (IkIndexError, offending value), or null without a list
    int *near;                 // [2] per list: problems whose last |Q_u|^2 was below kNearStop (the express lane's trigger statistics)
    int *xmeta, *xlist;        // the express lane: {taken, count, iteration, -} and the problems it took
    double near_stop;          // |Q_u|^2 below this counts as "within reach of the stopping threshold" in the lane's trigger statistics
    int iter, n_launch;
    const RobotModelDev *model;
    const double *x0;          // [B][37]
    const double *dt;          // [B][T]
    const double *tasks;       // [B][T+1][kNodeTaskDoubles]
    const double *state_w;     // [B or 1][36]   ActivationModelWeightedQuad weights of xReg
    const double *x_reg;       // [B][37]
    const double *ctrl_w;      // [B or 1][18]
    long s_state_w, s_ctrl_w;  // batch strides (0 = shared)
    long s_x_reg;              // batch stride of x_reg (37 for [B][37])
    long sn_state_w, sn_x_reg, sn_ctrl_w;   // node strides (0 = the same vector at every node of a problem)
    double *ws;                // [B][layout.total]
    int *active;               // device counter of problems still iterating
};

hipError_t ik_launch_init(const IkBatchArgs &a, hipStream_t s);
hipError_t ik_launch_state(const IkBatchArgs &a, hipStream_t s);      // before every calcdiff
hipError_t ik_launch_calcdiff(const IkBatchArgs &a, hipStream_t s);
hipError_t ik_launch_backward(const IkBatchArgs &a, hipStream_t s);
hipError_t ik_launch_forward(const IkBatchArgs &a, hipStream_t s);
// host_word_dev[0..3] = *active, index-check code, length of the list iteration next_iter runs over, express lane taken
// (device alias of four host-mapped ints)
// n_iters_of > 0: also host word [4] = the most DDP iterations any of the first n_iters_of problems ran (fused-direct path)
hipError_t ik_launch_publish_active(const IkBatchArgs &a, int next_iter, int *host_word_dev, hipStream_t s, int n_iters_of = 0);
hipError_t ik_launch_select(const IkBatchArgs &a, int cap, int force, hipStream_t s);   // force: tests (take the lane whatever the batch looks like)
hipError_t ik_launch_fused_express(const IkBatchArgs &a, int cap, hipStream_t side);
hipError_t ik_launch_fused_tail(const IkBatchArgs &a, hipStream_t s);
// centroidal state [com, vcom, L] (9) of (q, v): KinoDynMP::optimize's x0 (kino_dyn.cpp:42,86-97)
hipError_t ik_launch_centroidal_state(const RobotModelDev *model, const double *x, double *out9, int B, hipStream_t s);
// com (3) and h_g (6) along a state trajectory [B][n][37]  (InverseKinematics::return_opt_com/mom)
hipError_t ik_launch_com_mom(const RobotModelDev *model, const double *xs, double *com, double *mom, int n_states, hipStream_t s);

hipError_t ik_launch_state_ops_selftest(const double *x0, const double *x1, const double *dx, int n, double *dq, double *dr, double *iq, double *ir,
                                        hipStream_t s);
void ik_kernel_occupancy(int *out8);      // workgroups per CU of the IK kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor)
hipError_t ik_launch_fill_refs(double *tasks, const double *X, double m, int B, int H, int T, hipStream_t s);

}  // namespace bunmpc

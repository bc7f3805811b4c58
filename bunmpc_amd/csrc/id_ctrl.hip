// Output stage of the data path: the inverse-dynamics controller the reference evaluates once per
// millisecond of every rollout, for a batch of samples.
//   tau    = (rnea(q_des, v_des, a_des) - sum_j J_j^T [f_j; 0])[6:]          ISL/examples/controllers/robot_id_controller.py:57-81
//   tau_fb = -kp (q - q_des)[joints] - kd (v - v_des)[joints]                 :83
//   action = (tau + tau_fb + kd v[6:]) / kp + q[7:]   ("pd_target")           ISL/examples/iterative_algorithm/simulation.py:518-524
//   state  = [v | base_xy - foot_xy per end effector | q[2:]]                 simulation.py:156-175, 489-491
// pin.rnea / pin.computeFrameJacobian (pinocchio 2.6.9, absent here) are restated as in oracle/id_np.py: Newton-Euler in
// body frames with the foot forces entered as external forces on their bodies (which is what subtracting J^T f does),
// the free-flyer's rotation taken from the quaternion as given (no normalisation, Eigen's toRotationMatrix).
//
// Four lanes per sample, one per leg, the 3-body recursion of a leg in registers; 64 samples per 256-thread workgroup.
// Rows enter and leave through LDS with coalesced global accesses, which also makes every per-joint value addressable
// by a run-time index (a lane's joints are 3 leg + k).  A first version -- one sample per lane, the four legs unrolled --
// was 60 KB of code at one wave per SIMD and ran 3x slower.  ~3.3 kflop and ~1.2 KB of traffic per sample: at MI355X's
// 78.6 TFLOP/s fp64 vector and 8 TB/s the two bounds are within 2x of each other.
#include "id_types.h"
#include "rbd_quad.h"

namespace bunmpc {
namespace {
using namespace rbd;

constexpr double kGravity = 9.81;

RBD_D void quat_matrix(const double *q, double *R) {   // (x, y, z, w), as Eigen: not normalised
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// parent -> child for a motion (lin, ang): lin' = R^T (lin - p x ang), ang' = R^T ang
RBD_D void motion_to_child(const double *R, const double *p, const double *mo, double *o) {
    double t[3];
    cross3(p, mo + 3, t);
    UNROLL_RBD for (int c = 0; c < 3; ++c) t[c] = mo[c] - t[c];
    mat3Tvec(R, t, o);
    mat3Tvec(R, mo + 3, o + 3);
}

// Per-sample staging row in LDS (doubles; odd stride so that the rows spread over the banks):
//   phase 1:  q_des 0..18 | v_des 19..36 | a_des 37..54 | f 55..66 |            | tau 69..80
//   phase 2:  q     0..18 | v     19..36 | tau_fb 37..48 | action 49..60 | foot offsets 61..68 | tau 69..80
constexpr int kLd = 81, kColV = 19, kColA = 37, kColF = 55, kColFb = 37, kColAct = 49, kColRel = 61, kColTau = 69;
constexpr int kBlockSamples = 64, kBlockThreads = 4 * kBlockSamples;

struct IdLds {
    double rec[kMaxJoints + 1][16];     // RobotModelDev::rec
    double buf[kBlockSamples * kLd];
};

// rows [row0, row0 + 64) x W columns of a row-strided global array <-> LDS columns [col, col + W): consecutive threads
// touch consecutive addresses, instead of each lane striding through its own row (a cache line per lane and load)
template <int W>
RBD_D void stage_in(double *buf, int col, const double *g, long stride, long row0, long n, int tid) {
    for (int idx = tid; idx < kBlockSamples * W; idx += kBlockThreads) {
        const int r = idx / W, c = idx - r * W;
        if (row0 + r < n) buf[r * kLd + col + c] = g[(row0 + r) * stride + c];
    }
}
template <int W>
RBD_D void stage_out(const double *buf, int col, double *g, long row0, long n, int tid) {
    for (int idx = tid; idx < kBlockSamples * W; idx += kBlockThreads) {
        const int r = idx / W, c = idx - r * W;
        if (row0 + r < n) g[(row0 + r) * W + c] = buf[r * kLd + col + c];
    }
}

RBD_D void inertia_apply(const double *rc, const double *mo, double *f) {   // rc = packed body record
    double t[3];
    cross3(mo + 3, rc + 7, t);
    UNROLL_RBD for (int c = 0; c < 3; ++c) f[c] = rc[6] * (mo[c] + t[c]);
    cross3(rc + 7, f, t);
    f[3] = rc[10] * mo[3] + rc[11] * mo[4] + rc[12] * mo[5] + t[0];
    f[4] = rc[11] * mo[3] + rc[13] * mo[4] + rc[14] * mo[5] + t[1];
    f[5] = rc[12] * mo[3] + rc[14] * mo[4] + rc[15] * mo[5] + t[2];
}

RBD_D void leg_joint_rotation(const RobotModelDev &m, int i, const double *axis, double qi, double *R) {
    rodrigues(axis, qi, R);
    if (!m.R_identity[i]) mat3mul(m.R[i], R, R);      // URDF joint origins with rpy != 0: none in Solo12 / Go2
}

// Four lanes per sample, one per leg: only the joint rows are asked for, so each leg is an independent 3-body recursion
// off the base's velocity / acceleration -- no base wrench, no cross-leg state.
__global__ __launch_bounds__(kBlockThreads) void id_controller_kernel(const IdLaunch a) {
    __shared__ IdLds s;
    const int tid = threadIdx.x, leg = tid & 3;
    const long row0 = (long)blockIdx.x * kBlockSamples;
    const RobotModelDev &m = *a.model;
    const bmpc_id_batch_t &d = a.d;
    for (int i = tid; i < (kMaxJoints + 1) * 16; i += kBlockThreads) (&s.rec[0][0])[i] = (&m.rec[0][0])[i];
    stage_in<19>(s.buf, 0, d.q_des, d.s_q_des, row0, d.n, tid);
    stage_in<18>(s.buf, kColV, d.v_des, d.s_v_des, row0, d.n, tid);
    stage_in<18>(s.buf, kColA, d.a_des, d.s_a_des, row0, d.n, tid);
    stage_in<12>(s.buf, kColF, d.f, d.s_f, row0, d.n, tid);
    __syncthreads();
    double *row = s.buf + (tid >> 2) * kLd;
    const int lf = a.leg_foot[leg], lk = a.leg_foot_k[leg];
    const double fp[3] = {a.leg_foot_p[leg][0], a.leg_foot_p[leg][1], a.leg_foot_p[leg][2]};

    double qdj[kLegJoints], vdj[kLegJoints];
    {
        double Rb[9], V[6], A[6], R[kLegJoints][9], F[kLegJoints][6];
        const double quat[4] = {row[3], row[4], row[5], row[6]};
        quat_matrix(quat, Rb);
        UNROLL_RBD for (int c = 0; c < 6; ++c) { V[c] = row[kColV + c]; A[c] = row[kColA + c]; }
        A[0] += kGravity * Rb[6]; A[1] += kGravity * Rb[7]; A[2] += kGravity * Rb[8];   // R^T (0, 0, g)
        UNROLL_RBD for (int k = 0; k < kLegJoints; ++k) {
            const int i = kLegJoints * leg + k;
            double rc[16];
            UNROLL_RBD for (int c = 0; c < 16; ++c) rc[c] = s.rec[i + 1][c];       // p 0..2 | axis 3..5 | mass 6 | com 7..9 | I 10..15
            const double qi = row[7 + i], qdot = row[kColV + 6 + i], qddot = row[kColA + 6 + i];
            qdj[k] = qi; vdj[k] = qdot;
            leg_joint_rotation(m, i, rc + 3, qi, R[k]);
            double Vc[6], Ac[6], t[3], h[6];
            motion_to_child(R[k], rc, V, Vc);
            motion_to_child(R[k], rc, A, Ac);
            const double wJ[3] = {rc[3] * qdot, rc[4] * qdot, rc[5] * qdot};
            UNROLL_RBD for (int c = 0; c < 3; ++c) { Vc[3 + c] += wJ[c]; Ac[3 + c] += rc[3 + c] * qddot; }
            cross3(Vc, wJ, t);                                      // a += v x vJ,  vJ = (0, wJ)
            UNROLL_RBD for (int c = 0; c < 3; ++c) Ac[c] += t[c];
            cross3(Vc + 3, wJ, t);
            UNROLL_RBD for (int c = 0; c < 3; ++c) Ac[3 + c] += t[c];
            UNROLL_RBD for (int c = 0; c < 6; ++c) { V[c] = Vc[c]; A[c] = Ac[c]; }
            inertia_apply(rc, V, h);
            inertia_apply(rc, A, F[k]);                             // f = I a + v x* (I v)
            cross3(V + 3, h, t);
            UNROLL_RBD for (int c = 0; c < 3; ++c) F[k][c] += t[c];
            cross3(V + 3, h + 3, t);
            UNROLL_RBD for (int c = 0; c < 3; ++c) F[k][3 + c] += t[c];
            cross3(V, h, t);
            UNROLL_RBD for (int c = 0; c < 3; ++c) F[k][3 + c] += t[c];
        }
        {   // the contact force of this leg's end effector, turned into its body's frame, acts on that body
            const double fw[3] = {row[kColF + 3 * lf], row[kColF + 3 * lf + 1], row[kColF + 3 * lf + 2]};
            double fl[3], t[3];
            mat3Tvec(Rb, fw, fl);
            UNROLL_RBD for (int k = 0; k < kLegJoints; ++k) {
                mat3Tvec(R[k], fl, fl);
                cross3(fp, fl, t);
                const double on = lk == k ? 1.0 : 0.0;
                UNROLL_RBD for (int c = 0; c < 3; ++c) { F[k][c] -= on * fl[c]; F[k][3 + c] -= on * t[c]; }
            }
        }
        UNROLL_RBD for (int k = kLegJoints - 1; k >= 0; --k) {
            const int i = kLegJoints * leg + k;
            row[kColTau + i] = s.rec[i + 1][3] * F[k][3] + s.rec[i + 1][4] * F[k][4] + s.rec[i + 1][5] * F[k][5];
            if (k > 0) {
                double lin[3], ang[3], t[3];
                const double pk[3] = {s.rec[i + 1][0], s.rec[i + 1][1], s.rec[i + 1][2]};
                mat3vec(R[k], F[k], lin);
                mat3vec(R[k], F[k] + 3, ang);
                cross3(pk, lin, t);
                UNROLL_RBD for (int c = 0; c < 3; ++c) { F[k - 1][c] += lin[c]; F[k - 1][3 + c] += ang[c] + t[c]; }
            }
        }
    }
    __syncthreads();

    // measured state over the desired one; feedback, action, and the foot offset of the state row
    stage_in<19>(s.buf, 0, d.q, d.s_q, row0, d.n, tid);
    stage_in<18>(s.buf, kColV, d.v, d.s_v, row0, d.n, tid);
    __syncthreads();
    UNROLL_RBD for (int k = 0; k < kLegJoints; ++k) {
        const int i = kLegJoints * leg + k;
        const double qi = row[7 + i], vi = row[kColV + 6 + i], tau = row[kColTau + i];
        const double fb = -d.kp[i] * (qi - qdj[k]) - d.kd[i] * (vi - vdj[k]);
        row[kColFb + i] = fb;
        row[kColAct + i] = (tau + fb + d.kd[i] * vi) / d.kp[i] + qi;
    }
    if (d.state) {
        double Rb[9], x[3] = {fp[0], fp[1], fp[2]};
        const double quat[4] = {row[3], row[4], row[5], row[6]};
        quat_matrix(quat, Rb);
        UNROLL_RBD for (int k = kLegJoints - 1; k >= 0; --k) {
            const int i = kLegJoints * leg + k;
            if (lk < k) continue;
            double Rk[9], t[3];
            const double axis[3] = {s.rec[i + 1][3], s.rec[i + 1][4], s.rec[i + 1][5]};
            leg_joint_rotation(m, i, axis, row[7 + i], Rk);
            mat3vec(Rk, x, t);
            UNROLL_RBD for (int c = 0; c < 3; ++c) x[c] = s.rec[i + 1][c] + t[c];
        }
        double w[3];
        mat3vec(Rb, x, w);
        row[kColRel + 2 * lf] = row[0] - (row[0] + w[0]);
        row[kColRel + 2 * lf + 1] = row[1] - (row[1] + w[1]);
    }
    __syncthreads();
    if (d.tau_ff) stage_out<12>(s.buf, kColTau, d.tau_ff, row0, d.n, tid);
    if (d.tau_fb) stage_out<12>(s.buf, kColFb, d.tau_fb, row0, d.n, tid);
    if (d.action) stage_out<12>(s.buf, kColAct, d.action, row0, d.n, tid);
    if (d.state)      // [v (18) | offsets (8) | q[2:] (17)]: columns 19..36, 61..68, 2..18 of the staging row
        for (int idx = tid; idx < kBlockSamples * 43; idx += kBlockThreads) {
            const int r = idx / 43, c = idx - r * 43;
            const int src = c < 18 ? kColV + c : (c < 26 ? kColRel + c - 18 : c - 24);
            if (row0 + r < d.n) d.state[(row0 + r) * 43 + c] = s.buf[r * kLd + src];
        }
}

}  // namespace

const char *id_kernel_name() { return "id_controller_kernel"; }

int launch_id_batch(const IdLaunch &a, hipStream_t st) {
    hipLaunchKernelGGL(id_controller_kernel, dim3((unsigned)((a.d.n + kBlockSamples - 1) / kBlockSamples)), dim3(kBlockThreads), 0, st, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(BMPC_DEVICE_ERROR, std::string("inverse-dynamics kernel: ") + hipGetErrorString(e));
    return BMPC_OK;
}

}  // namespace bunmpc

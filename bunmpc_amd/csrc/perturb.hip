// Contact-conditioned perturbation of nominal states: the sampler of the data-collection loop,
// ISL/examples/iterative_algorithm/data_collection.py:188-262, for a batch of nominal states, on the device.
//   J       = LOCAL_WORLD_ALIGNED linear Jacobians of the feet the contact plan has on the ground          (:204-214)
//   d_pos   = (I - pinv(J) J) pos,   pos = mu + sigma z                                                     (:232-243)
//   d_vel   = (I - pinv(A) A) pos,   A = J * vel (each column scaled by its velocity draw: the reference's
//             elementwise `cnt_jac * perturbation_vel`; its Jdot term is zero, see oracle/perturb_np.py)   (:244-246)
//   q' = integrate(q, d_pos), v' = v + d_vel, drawn again while a foot of q' is below the ground           (:229, 249-259)
// I - pinv(M) M is the orthogonal projector onto the null space of M: rows are orthonormalised by modified Gram-Schmidt
// (two passes; rows that vanish against the others are dropped, which is what pinv's cut-off does to a rank deficiency)
// and the row-space component is subtracted -- no 18 x 18 matrices, no SVD.
// One thread per nominal state, walking its draws in order as the reference's while loop does: a few thousand states
// of ~50 kflop each, a launch that only has to keep the inputs of the next solve on the device; it is not a hot kernel
// and keeps its matrices in private memory.
#include "perturb_types.h"
#include "rbd_device.h"

namespace bunmpc {
namespace {
using namespace rbd;

constexpr int kMaxRows = 12;

// orthonormal basis of the row space of A (r x 18) -> Q (count x 18)
__device__ int row_space(const double (*A)[kNV], int r, double (*Q)[kNV]) {
    double maxn2 = 0.0;
    for (int i = 0; i < r; ++i) {
        double n2 = 0.0;
        for (int c = 0; c < kNV; ++c) n2 += A[i][c] * A[i][c];
        maxn2 = fmax(maxn2, n2);
    }
    int cnt = 0;
    for (int i = 0; i < r; ++i) {
        double w[kNV];
        for (int c = 0; c < kNV; ++c) w[c] = A[i][c];
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < cnt; ++j) {
                double d = 0.0;
                for (int c = 0; c < kNV; ++c) d += Q[j][c] * w[c];
                for (int c = 0; c < kNV; ++c) w[c] -= d * Q[j][c];
            }
        double n2 = 0.0;
        for (int c = 0; c < kNV; ++c) n2 += w[c] * w[c];
        if (n2 > 1e-26 * maxn2) {
            const double s = 1.0 / sqrt(n2);
            for (int c = 0; c < kNV; ++c) Q[cnt][c] = w[c] * s;
            ++cnt;
        }
    }
    return cnt;
}

__device__ void project_out(const double (*Q)[kNV], int cnt, const double *x, double *y) {
    for (int c = 0; c < kNV; ++c) y[c] = x[c];
    for (int j = 0; j < cnt; ++j) {
        double d = 0.0;
        for (int c = 0; c < kNV; ++c) d += Q[j][c] * x[c];
        for (int c = 0; c < kNV; ++c) y[c] -= d * Q[j][c];
    }
}

__global__ __launch_bounds__(64) void perturb_kernel(const PerturbLaunch a) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    const bmpc_perturb_batch_t &d = a.d;
    if (b >= d.B) return;
    const RobotModelDev &m = *a.model;
    double x[kNX], xn[kNX], dx[kNDX];
    for (int c = 0; c < kNQ; ++c) x[c] = d.q[(long)b * kNQ + c];
    for (int c = 0; c < kNV; ++c) x[kNQ + c] = d.v[(long)b * kNV + c];
    Kin k;
    kin_compute<false>(m, x, k);
    double J[kMaxRows][kNV], Qp[kMaxRows][kNV], Qv[kMaxRows][kNV];
    int r = 0;
    for (int e = 0; e < 4; ++e) {
        if (d.contact[(long)b * d.s_contact_b + (long)e * d.s_contact_e] != 1.0) continue;     // `== 1`, :200
        const int f = d.foot_frame[e], body = m.frame_body[f];
        double xf[3], t[3];
        frame_position(m, k, f, xf);
        for (int c = 0; c < kNV; ++c) {
            if (in_support(m, body, c)) {
                cross3(k.S[c] + 3, xf, t);
                for (int i = 0; i < 3; ++i) J[r + i][c] = k.S[c][i] + t[i];
            } else
                for (int i = 0; i < 3; ++i) J[r + i][c] = 0.0;
        }
        r += 3;
    }
    const int np = row_space(J, r, Qp);
    int chosen = -1;
    for (int draw = 0; draw < d.K && chosen < 0; ++draw) {
        const double *z = d.z + ((long)b * d.K + draw) * 36;
        double pos[kNV], vel[kNV];
        for (int c = 0; c < kNV; ++c) {
            const int g = c < 3 ? 0 : (c < 6 ? 1 : 2);
            pos[c] = d.mu[g] + d.sigma[g] * z[c];
            vel[c] = d.mu[3] + d.sigma[3] * z[kNV + c];
        }
        if (r == 0) {
            for (int c = 0; c < kNV; ++c) { dx[c] = pos[c]; dx[kNV + c] = vel[c]; }
        } else {
            project_out(Qp, np, pos, dx);
            double A[kMaxRows][kNV];
            for (int i = 0; i < r; ++i)
                for (int c = 0; c < kNV; ++c) A[i][c] = J[i][c] * vel[c];
            const int nvr = row_space(A, r, Qv);
            project_out(Qv, nvr, pos, dx + kNV);
        }
        state_integrate(x, dx, xn);
        kin_compute<false>(m, xn, k);
        bool ok = true;
        for (int e = 0; e < 4; ++e) {
            double xf[3];
            frame_position(m, k, d.foot_frame[e], xf);
            if (xf[2] < 0.0) ok = false;
        }
        if (ok) chosen = draw;
    }
    const double *src = chosen >= 0 ? xn : x;
    for (int c = 0; c < kNQ; ++c) d.q_out[(long)b * kNQ + c] = src[c];
    for (int c = 0; c < kNV; ++c) d.v_out[(long)b * kNV + c] = src[kNQ + c];
    d.chosen[b] = chosen;
}

}  // namespace

int launch_perturb(const PerturbLaunch &a, hipStream_t st) {
    hipLaunchKernelGGL(perturb_kernel, dim3((unsigned)((a.d.B + 63) / 64)), dim3(64), 0, st, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(BMPC_DEVICE_ERROR, std::string("perturbation kernel: ") + hipGetErrorString(e));
    return BMPC_OK;
}

}  // namespace bunmpc

// Device-side harness inputs (centroidal level): the contact plan and the centroidal cost references that
// SoloMpcGaitGen.create_cnt_plan / create_costs build per MPC call on the host
// (ISL/examples/mpc/abstract_cyclic_gen.py:159-414, 564-607; gait phase from ISL/src/gait_planner/gait_planner.cpp:41-58,
// 112-128), for a whole batch in HBM.  Same operations in the same order as bunmpc_amd/problems.py::contact_plan /
// centroidal_costs (the numpy restatement the tests compare with, bit for bit): fmod phases with the 1e-4 slack,
// round-half-even to 3 decimals, the Raibert / centrifugal step rule, the first-knot dt rule.
//
// One thread per (problem, foot) walks the knots (a contact location is copied from the previous knot while the foot
// stays in stance); one thread per problem integrates the nominal CoM.  Integer / branchy fp64 work, HBM-bound at
// (4E + 1 + 9) H doubles written per problem; nothing to stage.
#include <cmath>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/bunmpc.h"
#include "ik_types.h"
#include "rbd_quad.h"

// bit-for-bit agreement with the numpy restatement needs separate multiplies and adds (no fused contraction)
#pragma clang fp contract(off)

namespace bunmpc {
int set_error(int code, const std::string &msg);
namespace {

constexpr double kGravity = 9.81, kFootSize = 0.018;   // abstract_cyclic_gen.py:31,50

__device__ __forceinline__ double round3(double x) { return rint(x * 1000.0) / 1000.0; }   // np.round(x, 3)
__device__ __forceinline__ double round2(double x) { return rint(x * 100.0) / 100.0; }

__device__ __forceinline__ double gait_phi(double t, double period, double offset) { return fmod(t + offset * period, period); }
__device__ __forceinline__ double gait_phase(double t, double period, double sp, double offset) {   // gait_planner.cpp:46-58
    const double st = period * sp, phi = gait_phi(t, period, offset);
    return (phi <= st || fabs(phi - st) < 1e-4) ? 1.0 : 0.0;
}
__device__ __forceinline__ double gait_percent(double t, double period, double sp, double offset) {   // gait_planner.cpp:112-128
    const double st = period * sp, phi = gait_phi(t, period, offset);
    return phi <= st ? phi / st : (phi - st) / (period - st);
}

__global__ void plan_feet_kernel(const bmpc_plan_batch_t d) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)d.B * 4) return;
    const long b = id / 4;
    const int j = (int)(id % 4), H = d.n_col;
    const bmpc_gait_params_t &g = d.gaits[d.gait_id ? d.gait_id[b] : 0];
    const double sp = g.stance_percent[j], off = g.phase_offset[j], gdt = g.gait_dt, period = g.gait_period;
    const double t0 = d.t0[b], wdes = d.w_des[b];
    const double vx = d.v_des[b * 3], vy = d.v_des[b * 3 + 1];
    const double cx = round3(d.com[b * 3]), cy = round3(d.com[b * 3 + 1]), zh = d.com[b * 3 + 2];
    const double hx = d.hip_off ? d.hip_off[(b * 4 + j) * 2] : g.offsets_xy[j][0];
    const double hy = d.hip_off ? d.hip_off[(b * 4 + j) * 2 + 1] : g.offsets_xy[j][1];
    // np.cross(0.5 sqrt(z/g) vtrack, [0,0,w]) -> (a_y w, -a_x w)      (:285-286)
    const double kz = 0.5 * sqrt(zh / kGravity);
    const double angx = (kz * vy) * wdes, angy = -(kz * vx) * wdes;
    // raibert = 0.5 vtrack period sp - 0.05 (vtrack - v_des): vtrack = v_des on this path      (:279-281)
    const double rbx = 0.5 * vx * period * sp - 0.05 * (vx - vx), rby = 0.5 * vy * period * sp - 0.05 * (vy - vy);
    double *cp = d.cnt_plan + (b * H * 4 + j) * 4;     // stride 16 doubles per knot
    double *sw = d.swing_time + b * H * 4 + j;         // stride 4 per knot
    double pflag = gait_phase(t0, period, sp, off);
    double px = round3(d.feet0[(b * 4 + j) * 3]), py = round3(d.feet0[(b * 4 + j) * 3 + 1]), pz = round3(d.feet0[(b * 4 + j) * 3 + 2]);
    cp[0] = pflag; cp[1] = px; cp[2] = py; cp[3] = pz;
    sw[0] = 0.0;
    for (int i = 1; i < H; ++i) {
        const double ft = round3(t0 + i * gdt);
        const double ph = gait_phase(ft, period, sp, off);
        const double hipx = cx + hx + i * gdt * vx, hipy = cy + hy + i * gdt * vy;
        const double per = round3(gait_percent(ft, period, sp, off));
        double x, y, z, s = 0.0;
        if (ph == 1.0) {
            if (pflag == 1.0) { x = px; y = py; z = pz; }
            else { x = rbx + hipx + angx; y = rby + hipy + angy; z = kFootSize; }
        } else {
            if (per < 0.5) { x = hipx + angx; y = hipy + angy; }
            else { x = hipx + angx + rbx; y = hipy + angy + rby; }
            z = kFootSize;
            s = (per - 0.5 < 0.02) ? 1.0 : 0.0;
        }
        double *c = cp + (long)i * 16;
        c[0] = ph; c[1] = x; c[2] = y; c[3] = z;
        sw[(long)i * 4] = s;
        pflag = ph; px = x; py = y; pz = z;
    }
}

__global__ void plan_costs_kernel(const bmpc_plan_batch_t d) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    const int H = d.n_col;
    const bmpc_gait_params_t &g = d.gaits[d.gait_id ? d.gait_id[b] : 0];
    const double gdt = g.gait_dt;
    double *dt = d.dt + b * H;
    {   // first-knot rule (:385-388)
        const double t0 = d.t0[b];
        const double r = fmod(t0, gdt);                       // np.remainder for non-negative operands
        const double d0 = gdt - round2(r);
        dt[0] = d0 == 0.0 ? gdt : d0;
        for (int i = 1; i < H; ++i) dt[i] = gdt;
    }
    const double *x0 = d.x_init + b * 9, *vd = d.v_des + b * 3;
    const double am[3] = {d.amom ? d.amom[b * 3] : 0.0, d.amom ? d.amom[b * 3 + 1] : 0.0, d.amom ? d.amom[b * 3 + 2] : 0.0};
    double *Xn = d.X_nom + b * 9L * H, *Xt = d.X_ter + b * 9;
    double nx = x0[0], ny = 0.0;                               // X_nom[0::9] = X_init[0]; y starts at 0 (:574-577)
    for (int i = 0; i < H; ++i) {
        if (i > 0) { nx = nx + vd[0] * dt[i]; ny = ny + vd[1] * dt[i]; }
        double *r = Xn + 9L * i;
        r[0] = i == 0 ? x0[0] : nx; r[1] = i == 0 ? 0.0 : ny; r[2] = g.nom_ht;
        r[3] = vd[0]; r[4] = vd[1]; r[5] = vd[2];
        r[6] = am[0] * g.ori_correction[0]; r[7] = am[1] * g.ori_correction[1]; r[8] = am[2] * g.ori_correction[2];
    }
    Xt[0] = x0[0] + g.gait_horizon * g.gait_period * vd[0];
    Xt[1] = x0[1] + g.gait_horizon * g.gait_period * vd[1];
    Xt[2] = g.nom_ht; Xt[3] = vd[0]; Xt[4] = vd[1]; Xt[5] = vd[2];
    Xt[6] = am[0]; Xt[7] = am[1]; Xt[8] = am[2];
}

// whole-body front end: kinematics of x = [q, v] and the quantities the plan builders take
__global__ void wb_state_kernel(const RobotModelDev *model, const bmpc_wb_plan_batch_t d) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    const double *x = d.x + b * kNX;
    rbd::Pass1 p1;
    const int fid[kFrameSlots] = {d.foot_frame[0], d.foot_frame[1], d.foot_frame[2], d.foot_frame[3]};
    rbd::quad_pass1<false>(*model, x, fid, p1);
    for (int c = 0; c < 3; ++c) {
        d.com[b * 3 + c] = p1.com[c];
        d.x_init[b * 9 + c] = p1.com[c];
        d.x_init[b * 9 + 3 + c] = p1.hg[c] / p1.M;
        d.x_init[b * 9 + 6 + c] = p1.hg[3 + c];
    }
    for (int j = 0; j < 4; ++j) for (int c = 0; c < 3; ++c) d.feet0[(b * 4 + j) * 3 + c] = p1.fx[j][c];
    const double *vb = d.v_des_body + b * 3;
    for (int c = 0; c < 3; ++c) d.v_des[b * 3 + c] = p1.Rb[3 * c] * vb[0] + p1.Rb[3 * c + 1] * vb[1] + p1.Rb[3 * c + 2] * vb[2];   // :642-643
    d.w_des[b] = 0.0;
    const double yaw = atan2(p1.Rb[3], p1.Rb[0]), cy = cos(yaw), sy = sin(yaw);    // matrixToRpy, roll = pitch = 0 (:173-177)
    for (int j = 0; j < 4; ++j) {
        const double ox = d.gait->offsets_xy[j][0], oy = d.gait->offsets_xy[j][1];
        d.hip_off[(b * 4 + j) * 2] = cy * ox - sy * oy;
        d.hip_off[(b * 4 + j) * 2 + 1] = sy * ox + cy * oy;
    }
    double Rt[9], w[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = p1.Rb[3 * j + i];
    rbd::log3(Rt, w);                                                              // log3(R_des R_q^T), R_des = I (:616-627)
    for (int c = 0; c < 3; ++c) d.amom[b * 3 + c] = w[c];
}

// IK task blocks (abstract_cyclic_gen.py:545-562): 4 x {w, frame, ref3} | com {w, ref3} | mom {w, ref6} | state w | ctrl w
__global__ void wb_tasks_kernel(const bmpc_wb_plan_batch_t d) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nn = d.ik_col + 1;
    if (id >= (long)d.B * nn) return;
    const long b = id / nn;
    const int t = (int)(id % nn);
    double *tk = d.ik_tasks + id * kNodeTaskDoubles;
    for (int i = 0; i < kNodeTaskDoubles; ++i) tk[i] = 0.0;
    if (t < d.ik_col) {
        for (int j = 0; j < 4; ++j) {
            const double *c = d.cnt_plan + ((b * d.n_col + t) * 4 + j) * 4;
            const bool on = c[0] == 1.0, via = !on && d.swing_time[(b * d.n_col + t) * 4 + j] == 1.0;
            tk[5 * j] = on ? d.swing_wt[0] : (via ? d.swing_wt[1] : 0.0);
            tk[5 * j + 1] = (double)d.foot_frame[j];
            tk[5 * j + 2] = c[1]; tk[5 * j + 3] = c[2]; tk[5 * j + 4] = via ? d.step_ht : c[3];
        }
    }
    tk[5 * kFrameSlots] = d.cent_wt[0];
    tk[5 * kFrameSlots + 4] = d.cent_wt[1];
    tk[5 * kFrameSlots + 11] = d.reg_wt[0];
    tk[5 * kFrameSlots + 12] = d.reg_wt[1];
}

// 1 kHz resampling of the first `size` knot intervals of a planned trajectory (abstract_cyclic_gen.py:677-692):
// vstack_i linspace(knots[i], knots[i+1], int(dt_i / step)) -- end points included and therefore repeated at the
// seams, as there.  One thread per (problem, output row, component); rows beyond a problem's total are left alone.
__global__ void interp_kernel(const bmpc_interp_batch_t d) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)d.max_rows * d.width;
    if (id >= (long)d.B * per) return;
    const long b = id / per;
    const int row = (int)((id % per) / d.width), c = (int)(id % d.width);
    const double *dt = d.dt + b * d.dt_stride;
    int i = 0, first = 0, n = 0;
    for (; i < d.size; ++i) {                      // which interval holds this row
        n = (int)(dt[i] / d.step);                 // int(dt_arr[i] / 0.001)
        if (row < first + n) break;
        first += n;
    }
    if (c == 0 && row == 0) {
        int tot = 0;
        for (int k = 0; k < d.size; ++k) tot += (int)(dt[k] / d.step);
        d.rows[b] = tot;
    }
    if (i == d.size) return;
    const int k = row - first;
    const double a0 = d.knots[(b * d.n_knots + i) * d.width + c], a1 = d.knots[(b * d.n_knots + i + 1) * d.width + c];
    double y;
    if (n == 1) y = a0;
    else if (k == n - 1) y = a1;                   // numpy.linspace sets the end point exactly
    else {
        const double stepv = (a1 - a0) / (double)(n - 1);
        y = stepv == 0.0 ? ((double)k / (double)(n - 1)) * (a1 - a0) + a0 : (double)k * stepv + a0;
    }
    d.out[(b * d.max_rows + row) * d.width + c] = y;
}

}  // namespace

int launch_wb_plan(const RobotModelDev *model, const bmpc_wb_plan_batch_t &d, hipStream_t st) {
    hipLaunchKernelGGL(wb_state_kernel, dim3((unsigned)((d.B + 63) / 64)), dim3(64), 0, st, model, d);
    bmpc_plan_batch_t p;
    p.B = d.B; p.n_col = d.n_col; p.n_gaits = 1; p.reserved_ = 0;
    p.gaits = d.gait; p.gait_id = nullptr;
    p.t0 = d.t0; p.com = d.com; p.feet0 = d.feet0; p.v_des = d.v_des; p.w_des = d.w_des; p.x_init = d.x_init;
    p.amom = d.amom; p.hip_off = d.hip_off;
    p.cnt_plan = d.cnt_plan; p.swing_time = d.swing_time; p.dt = d.dt; p.X_nom = d.X_nom; p.X_ter = d.X_ter;
    const long nf = (long)d.B * 4;
    hipLaunchKernelGGL(plan_feet_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(plan_costs_kernel, dim3((unsigned)((d.B + 255) / 256)), dim3(256), 0, st, p);
    const long nt = (long)d.B * (d.ik_col + 1);
    hipLaunchKernelGGL(wb_tasks_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, d);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(BMPC_DEVICE_ERROR, std::string("whole-body plan kernels: ") + hipGetErrorString(e));
    return BMPC_OK;
}

}  // namespace bunmpc

extern "C" int bmpc_interp_batch_device(const bmpc_interp_batch_t *d, void *hip_stream) {
    using namespace bunmpc;
    if (!d || !d->knots || !d->dt || !d->out || !d->rows) return set_error(BMPC_BAD_ARG, "null interpolation argument");
    if (d->B < 0 || d->size < 1 || d->size >= d->n_knots || d->width < 1 || d->max_rows < 1 || !(d->step > 0.0))
        return set_error(BMPC_BAD_ARG, "bad interpolation sizes");
    if (d->B == 0) return BMPC_OK;
    const long n = (long)d->B * d->max_rows * d->width;
    hipLaunchKernelGGL(interp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), *d);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(BMPC_DEVICE_ERROR, std::string("interpolation kernel: ") + hipGetErrorString(e));
    return BMPC_OK;
}

extern "C" int bmpc_plan_batch_device(const bmpc_plan_batch_t *d, void *hip_stream) {
    using namespace bunmpc;
    if (!d) return set_error(BMPC_BAD_ARG, "null plan descriptor");
    if (d->B < 0 || d->n_col < 1) return set_error(BMPC_BAD_ARG, "B < 0 or n_col < 1");
    if (!d->gaits || d->n_gaits < 1) return set_error(BMPC_BAD_ARG, "no gait parameters");
    if (d->n_gaits > 1 && !d->gait_id) return set_error(BMPC_BAD_ARG, "several gaits need gait_id");
    if (!d->t0 || !d->com || !d->feet0 || !d->v_des || !d->w_des || !d->x_init) return set_error(BMPC_BAD_ARG, "missing input array");
    if (!d->cnt_plan || !d->swing_time || !d->dt || !d->X_nom || !d->X_ter) return set_error(BMPC_BAD_ARG, "missing output array");
    if (d->B == 0) return BMPC_OK;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const long nf = (long)d->B * 4;
    hipLaunchKernelGGL(plan_feet_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, *d);
    hipLaunchKernelGGL(plan_costs_kernel, dim3((unsigned)((d->B + 255) / 256)), dim3(256), 0, st, *d);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(BMPC_DEVICE_ERROR, std::string("plan kernels: ") + hipGetErrorString(e));
    return BMPC_OK;
}

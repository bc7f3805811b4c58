// Centroidal bi-convex ADMM, ONE PROBLEM PER WAVE (gfx950): the mapping for small batches and short horizons, where the
// time of a solve is the length of its dependent instruction chain and not the number of problems in flight.  Same
// algorithm, same algebra and the same reference lines as biconvex_admm.hip (see the header there); what differs is how
// one problem is spread over the 64 lanes:
//   * force step   lane = 32 h + t : knot t (t < H <= 20), feet 2h and 2h + 1.  The 6 x 12 block of A_x of a knot is the sum
//                  of the two halves' 6 x 6 blocks: v_permlane32_swap adds them, after which both lanes of a knot hold the
//                  same six residual rows.
//   * motion step  lane = 21 g + t : knot t (t <= H), component group g = CoM / velocity / angular momentum (three of the
//                  nine components each; lane 63 idles).  Knot t +- 1 of the same group sits in the neighbouring lane (DPP
//                  wave shifts, as in the batch kernel; the lane before a group's first knot is the previous group's knot 20,
//                  whose row block does not exist and is held at zero, which is exactly what row block -1 must read as);
//                  the few entries of A_f that couple groups -- dt v_{t+1} in the CoM rows, S x com_t in the momentum rows,
//                  and their transposes -- fetch three doubles from one other lane through ds_bpermute and enter as a
//                  per-lane 3 x 3 pattern a f_k + b f_{k+1} + c f_{k+2} whose coefficients are zero where a group has no
//                  such term.
// A wave therefore executes 6 of the 12 force components and 3 of the 9 state components per instruction stream instead of
// all of them: ~2.3x fewer vector instructions per FISTA iteration than one knot per lane, at 40 / 63 busy lanes.  With one
// problem per wave every decision is wave-uniform (scalar branches, no lane masks), the iterates of the problem rest in LDS
// between phases, and nothing crosses waves.
//
// Results: the same discrete path as the batch kernel and the CPU oracle (iteration and retry counts), values equal to
// rounding (the segment sums run over another lane order): tests/test_biconvex_gpu.py::test_latency_mapping_*.
#include "biconvex_kernels.h"

namespace bunmpc {
namespace {

#include "biconvex_lanes.h"

constexpr int kLatKnots = 21;            // H + 1 <= 21: three groups of 21 lanes

__device__ __forceinline__ double bperm(double v, int src_lane_bytes) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane_bytes, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane_bytes, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <bool RAW, bool HASQF>
__global__ __launch_bounds__(64) void biconvex_latency_kernel(const BatchArgs a) {
    extern __shared__ double lds_raw[];
    const int lane = threadIdx.x;
    const int H = a.H, maxit = a.c.maxit;
    const long pb = blockIdx.x;
    const long nx = 9L * (H + 1), nf = 12L * H;
    double *cmtab = lds_raw;                                   // [maxit]
    double *Xs = cmtab + ((maxit + 1) & ~1), *Ps = Xs + nx, *Fs = Ps + nx;     // iterates of the problem between phases
    const double m = a.c.m, rho = a.c.rho, mu = a.c.mu, beta = a.c.beta, tol = a.c.tol, exit_tol = a.c.exit_tol;
    const double tol2 = tol * tol;

    // force-step role
    const int fh = lane >> 5, ft = lane & 31;
    const bool fvalid = ft < H;
    // motion-step role
    const int xg = lane / kLatKnots, xt = lane % kLatKnots;
    const bool kvalid = xg < 3 && xt <= H, rvalid = xg < 3 && xt < H, x0lane = xg < 3 && xt == 0;

    {   // momentum table: t+ = 1 + sqrt(1 + 4 t^2)/2 (sic, fista.cpp:34), c = (t - 1)/t+
        double tk = 1.0;
        for (int i = 0; i < maxit; ++i) {
            const double tk1 = 1.0 + sqrt(1.0 + 4.0 * tk * tk) * 0.5;
            if (lane == 0) cmtab[i] = (tk - 1.0) / tk1;
            tk = tk1;
        }
    }
    const bool fresh_L = a.cold_start == 1;
    double L_x = fresh_L ? a.L0_x : a.L_x[pb], L_f = fresh_L ? a.L0_f : a.L_f[pb];
    if (a.cold_start != 0) {     // KinoDynMP::set_warm_starts (kino_dyn.cpp:83-99): X = tile(x_init), F = 0, P = 0
        for (long i = lane; i < nx; i += 64) { Xs[i] = a.x_init[pb * 9 + i % 9]; Ps[i] = 0.0; }
        for (long i = lane; i < nf; i += 64) Fs[i] = 0.0;
    } else {
        for (long i = lane; i < nx; i += 64) { Xs[i] = a.X[pb * nx + i]; Ps[i] = a.P[pb * nx + i]; }
        for (long i = lane; i < nf; i += 64) Fs[i] = a.F[pb * nf + i];
    }
    __syncthreads();

    bool alive = true;
    int n_admm = 0, it_f = 0, it_x = 0, bt_f = 0, bt_x = 0, status = 0;
    double last_viol = 0.0;

    for (int it = 0; it < a.c.num_iters && alive; ++it) {
        // =================================================================== F step: lane = (half fh, knot ft)
        {
            const double *cg = a.cnt_plan + ((pb * H + (fvalid ? ft : 0)) * 4 + 2 * fh) * 4;     // feet 2 fh, 2 fh + 1 of the knot
            const double dt = fvalid ? a.dt[pb * H + ft] : 0.0;
            double an[2], sp[2][3], wf[6], qf[HASQF ? 6 : 1];
            UNROLL for (int n = 0; n < 2; ++n) {
                const double c = fvalid ? cg[4 * n] : 0.0;
                an[n] = c * (dt / m);
                UNROLL for (int k = 0; k < 3; ++k) sp[n][k] = fvalid ? c * (Xs[9 * ft + k] - cg[4 * n + 1 + k]) * dt : 0.0;
            }
            const long fo = 12L * ft + 6 * fh;
            UNROLL for (int j = 0; j < 6; ++j) {
                wf[j] = fvalid ? (RAW ? a.Qf[pb * nf + fo + j] : a.W_F[pb * a.sW_F + fo + j]) : 0.0;
                if (HASQF) qf[j] = fvalid ? 0.5 * a.qf[pb * nf + fo + j] : 0.0;
            }
            // bPk rows 9t+3..8 = -b_x + P, b_x = X_{t+1} - X_t (+g dt)   (centroidal.cpp:60-65)
            double bpk[6];
            UNROLL for (int k = 0; k < 6; ++k) {
                double bx = fvalid ? Xs[9 * (ft + 1) + 3 + k] - Xs[9 * ft + 3 + k] : 0.0;
                if (k == 2) bx += kGravity * dt;
                bpk[k] = fvalid ? (-bx + Ps[9 * ft + 3 + k]) : 0.0;
            }
            // u = A v + bPk on rows 9t+3..8: this half's two feet, then the other half's share
            auto applyA = [&](const double (&v)[6], double (&u)[6]) {
                double s[6] = {0, 0, 0, 0, 0, 0};
                UNROLL for (int n = 0; n < 2; ++n) {
                    const double vx = v[3 * n], vy = v[3 * n + 1], vz = v[3 * n + 2];
                    s[0] += an[n] * vx; s[1] += an[n] * vy; s[2] += an[n] * vz;
                    s[3] += sp[n][2] * vy - sp[n][1] * vz;
                    s[4] += sp[n][0] * vz - sp[n][2] * vx;
                    s[5] += sp[n][1] * vx - sp[n][0] * vy;
                }
                UNROLL for (int k = 0; k < 6; ++k) u[k] = swap32_sum(s[k]) + bpk[k];
            };
            double xa[6], xb[6], y[6], ra[6], rb[6], ry[6];
            UNROLL for (int j = 0; j < 6; ++j) { xa[j] = fvalid ? Fs[fo + j] : 0.0; y[j] = xa[j]; }
            applyA(y, ry);
            UNROLL for (int k = 0; k < 6; ++k) ra[k] = ry[k];
            const double mu2 = mu * mu, imu = 1.0 / (mu * mu + 1.0);
            // the fp32 shortcut of the step decisions (banded_decisions) presumes sums of non-negative terms
            const bool banded_f = !a.exact_step_decisions && rho >= 0.0 && !__any(wf[0] < 0 || wf[1] < 0 || wf[2] < 0 || wf[3] < 0 || wf[4] < 0 || wf[5] < 0);
            double invL = 2.0 * (1.0 / L_f);      // the gradient is carried as half of itself (biconvex_admm.hip)
            const double e2w = fh == 0 ? rho : 0.0;   // the residual rows are held twice: counted once
            bool act = true;
            auto iterate = [&](const double (&xo)[6], const double (&ro)[6], double (&xn)[6], double (&rn)[6], int i) {
                const double cm = cmtab[i];
                bool done;
                for (;;) {      // backtracking (fista.cpp:8-26)
                    double fr[6];
                    bool cone_any = false;
                    UNROLL for (int n = 0; n < 2; ++n) {
                        const double zx = an[n] * ry[0] - sp[n][2] * ry[4] + sp[n][1] * ry[5];
                        const double zy = an[n] * ry[1] + sp[n][2] * ry[3] - sp[n][0] * ry[5];
                        const double zz = an[n] * ry[2] - sp[n][1] * ry[3] + sp[n][0] * ry[4];
                        double gx = fma(wf[3 * n], y[3 * n], rho * zx), gy = fma(wf[3 * n + 1], y[3 * n + 1], rho * zy),
                               gz = fma(wf[3 * n + 2], y[3 * n + 2], rho * zz);
                        if (HASQF) { gx += qf[3 * n]; gy += qf[3 * n + 1]; gz += qf[3 * n + 2]; }
                        fr[3 * n] = fma(-gx, invL, y[3 * n]);
                        fr[3 * n + 1] = fma(-gy, invL, y[3 * n + 1]);
                        fr[3 * n + 2] = fma(-gz, invL, y[3 * n + 2]);
                        // "SoC" projection exactly as fista.cpp:52-70 writes it
                        const double s = fma(fr[3 * n], fr[3 * n], fr[3 * n + 1] * fr[3 * n + 1]);
                        const double fz = fr[3 * n + 2];
                        const bool zero = (s * mu < -fz) || (fz < 0);
                        cone_any = cone_any || (!zero && (s > mu * fz));
                        const double keep = zero ? 0.0 : 1.0;
                        xn[3 * n] = keep * fr[3 * n];
                        xn[3 * n + 1] = keep * fr[3 * n + 1];
                        xn[3 * n + 2] = keep * fz;
                    }
                    if (__any(cone_any)) {   // cone branch (fista.cpp:64-68); skipped while no lane needs it
                        UNROLL for (int n = 0; n < 2; ++n) {
                            const double s = fma(fr[3 * n], fr[3 * n], fr[3 * n + 1] * fr[3 * n + 1]);
                            const double fz = fr[3 * n + 2];
                            const bool zero = (s * mu < -fz) || (fz < 0);
                            const bool cone = !zero && (s > mu * fz);
                            const double k = fast_div(fma(mu2, s, mu * fz), (mu2 + 1.0) * s);
                            xn[3 * n] = cone ? fr[3 * n] * k : xn[3 * n];
                            xn[3 * n + 1] = cone ? fr[3 * n + 1] * k : xn[3 * n + 1];
                            xn[3 * n + 2] = cone ? fma(mu, s, fz) * imu : xn[3 * n + 2];
                        }
                    }
                    applyA(xn, rn);
                    double g2 = 0, cv = 0, e2 = 0;
                    UNROLL for (int j = 0; j < 6; ++j) {
                        const double d = xn[j] - y[j];
                        g2 = fma(d, d, g2);
                        cv = fma(wf[j] * d, d, cv);
                    }
                    UNROLL for (int k = 0; k < 6; ++k) { const double e = rn[k] - ry[k]; e2 = fma(e, e, e2); }
                    cv = fma(e2w, e2, cv);
                    const double Lh = L_f * 0.5;
                    bool bt;
                    if (!(banded_f && banded_decisions(g2, cv, Lh, tol2, bt, done))) {
                        double g2s = g2, cvs = cv;
                        seg_sum2<64>(g2s, cvs);
                        const double rhs = Lh * g2s;     // fista.cpp:14-17, sqrt only where it could matter (biconvex_admm.hip)
                        bt = cvs > rhs;
                        done = g2s < tol2;
                        if ((fabs(cvs - rhs) <= 1e-14 * rhs) || (fabs(g2s - tol2) <= 1e-14 * tol2)) {
                            const double Gn = sqrt(g2s);
                            bt = cvs > Lh * (Gn * Gn);
                            done = Gn < tol;
                        }
                    }
                    if (!__any(bt)) break;
                    L_f *= beta; ++bt_f;
                    invL = 2.0 * (1.0 / L_f);
                }
                done = __any(done);
                if ((done || i == maxit - 1) && fvalid) { UNROLL for (int j = 0; j < 6; ++j) Fs[fo + j] = xn[j]; }
                UNROLL for (int j = 0; j < 6; ++j) y[j] = fma(cm, xn[j] - xo[j], xn[j]);
                UNROLL for (int k = 0; k < 6; ++k) ry[k] = fma(cm, rn[k] - ro[k], rn[k]);
                ++it_f;
                act = !done;
            };
            for (int i = 0; i < maxit && act; i += 2) {
                iterate(xa, ra, xb, rb, i);
                if (i + 1 >= maxit || !act) break;
                iterate(xb, rb, xa, ra, i + 1);
            }
        }
        __syncthreads();

        // =================================================================== X step: lane = (group xg, knot xt)
        {
            const int tq = rvalid ? xt : 0;
            const double *cg = a.cnt_plan + (pb * H + tq) * 16;
            const double dt = rvalid ? a.dt[pb * H + xt] : 0.0;
            const double dtp = (xg < 3 && xt >= 1 && xt <= H) ? a.dt[pb * H + xt - 1] : 0.0;
            double c[4], r[4][3];
            UNROLL for (int n = 0; n < 4; ++n) {
                c[n] = rvalid ? cg[4 * n] : 0.0;
                UNROLL for (int k = 0; k < 3; ++k) r[n][k] = rvalid ? cg[4 * n + 1 + k] : 0.0;
            }
            // A_f / b_f entries of this knot from the new forces (centroidal.cpp:86-127)
            double SX = 0, SY = 0, SZ = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0;
            UNROLL for (int n = 0; n < 4; ++n) {
                const double fx = rvalid ? Fs[12 * xt + 3 * n] : 0.0, fy = rvalid ? Fs[12 * xt + 3 * n + 1] : 0.0,
                             fz = rvalid ? Fs[12 * xt + 3 * n + 2] : 0.0;
                SX += c[n] * fx * dt; SY += c[n] * fy * dt; SZ += c[n] * fz * dt;
                b3 += -c[n] * fx * dt / m; b4 += -c[n] * fy * dt / m; b5 += -c[n] * fz * dt / m;
                b6 += (c[n] * fy * r[n][2] - c[n] * fz * r[n][1]) * dt;
                b7 += (c[n] * fz * r[n][0] - c[n] * fx * r[n][2]) * dt;
                b8 += (c[n] * fx * r[n][1] - c[n] * fy * r[n][0]) * dt;
            }
            b5 += kGravity * dt;
            const double bf[3] = {xg == 1 ? b3 : xg == 2 ? b6 : 0.0, xg == 1 ? b4 : xg == 2 ? b7 : 0.0, xg == 1 ? b5 : xg == 2 ? b8 : 0.0};
            const int co = 9 * xt + 3 * (xg < 3 ? xg : 0);          // this lane's three components in X / P
            double bpk[3];
            UNROLL for (int k = 0; k < 3; ++k) bpk[k] = rvalid ? (-bf[k] + Ps[co + k]) : 0.0;
            // cost and bounds of this lane's components
            double qd[3], q[3], lb[3], ub[3];
            if (RAW) {
                UNROLL for (int k = 0; k < 3; ++k) {
                    qd[k] = kvalid ? a.Qx[pb * nx + co + k] : 0.0;
                    q[k] = kvalid ? 0.5 * a.qx[pb * nx + co + k] : 0.0;
                    lb[k] = kvalid ? a.lbx[pb * nx + co + k] : -INFINITY;
                    ub[k] = kvalid ? a.ubx[pb * nx + co + k] : INFINITY;
                }
            } else {
                UNROLL for (int k = 0; k < 3; ++k) {      // create_cost_X (biconvex.cpp:57-72)
                    const int l = 3 * (xg < 3 ? xg : 0) + k;
                    const double w = rvalid ? a.W_X[pb * a.sW_X + 9L * xt + l] : (kvalid ? a.W_X_ter[pb * a.sW_X_ter + l] : 0.0);
                    const double xr = rvalid ? a.X_nom[pb * 9L * H + 9L * xt + l] : (kvalid ? a.X_ter[pb * 9 + l] : 0.0);
                    qd[k] = w;
                    q[k] = -(xr * w);
                }
                // create_bound_constraints (biconvex.cpp:27-55): CoM box around the feet
                double csum = 0;
                UNROLL for (int n = 0; n < 4; ++n) csum += c[n];
                const bool bounded = rvalid && xg == 0 && csum > 0;
                UNROLL for (int k = 0; k < 3; ++k) {
                    double mx = r[0][k], mn = r[0][k];
                    UNROLL for (int n = 1; n < 4; ++n) { mx = fmax(mx, r[n][k]); mn = fmin(mn, r[n][k]); }
                    lb[k] = bounded ? mx + a.bounds[pb * a.sbounds + 6L * xt + k] : -INFINITY;
                    ub[k] = bounded ? mn + a.bounds[pb * a.sbounds + 6L * xt + 3 + k] : INFINITY;
                }
            }
            // x_init rows folded into knot 0's diagonal cost:  rho |X_0 + (P_H - x_init)|^2
            UNROLL for (int k = 0; k < 3; ++k) {
                const double bpi = x0lane ? (Ps[9 * H + 3 * xg + k] - a.x_init[pb * 9 + 3 * xg + k]) : 0.0;
                qd[k] += x0lane ? rho : 0.0;
                q[k] = fma(rho, bpi, q[k]);
                lb[k] = __builtin_canonicalize(lb[k]);
                ub[k] = __builtin_canonicalize(ub[k]);
            }
            // the entries of A_f (and of its transpose) that couple the groups: u_k += a f_k + b f_{k+1} + c f_{k+2}, f from one other lane
            //   A_f:    CoM rows       += dt v_{t+1}        (f = velocity group of knot t + 1: lane + 22)
            //           momentum rows  += S x com_t         (f = CoM group of knot t: lane - 42)
            //   A_f^T:  CoM entries    += (S x . )^T r_L,t  (f = momentum rows of knot t: lane + 42)
            //           velocity ent.  += dt_{t-1} r_com,t-1 (f = CoM rows of knot t - 1: lane - 22)
            double a1[3], b1[3], c1[3], a2[3], b2[3], c2[3];
            UNROLL for (int k = 0; k < 3; ++k) { a1[k] = b1[k] = c1[k] = a2[k] = b2[k] = c2[k] = 0.0; }
            int src1 = lane, src2 = lane;
            if (xg == 0) {
                a1[0] = a1[1] = a1[2] = dt; src1 = lane + 22 < 64 ? lane + 22 : lane;
                b2[0] = SZ; c2[0] = -SY; b2[1] = SX; c2[1] = -SZ; b2[2] = SY; c2[2] = -SX; src2 = lane + 42 < 64 ? lane + 42 : lane;
            } else if (xg == 1) {
                a2[0] = a2[1] = a2[2] = dtp; src2 = xt >= 1 ? lane - 22 : lane;
            } else if (xg == 2) {
                b1[0] = -SZ; c1[0] = SY; b1[1] = -SX; c1[1] = SZ; b1[2] = -SY; c1[2] = SX; src1 = lane - 42;
            }
            src1 *= 4; src2 *= 4;
            const int rmask = rvalid ? -1 : 0;
            // u = A_f v + bPk on this lane's three rows of row-block t
            auto applyA0 = [&](const double (&v)[3], double (&w)[3]) {
                double f[3];
                UNROLL for (int k = 0; k < 3; ++k) f[k] = bperm(v[k], src1);
                UNROLL for (int k = 0; k < 3; ++k) {
                    const double vn = from_next(v[k]);
                    w[k] = v[k] - vn;
                }
                UNROLL for (int k = 0; k < 3; ++k) w[k] += fma(a1[k], f[k], fma(b1[k], f[(k + 1) % 3], c1[k] * f[(k + 2) % 3]));
            };
            auto applyA = [&](const double (&v)[3], double (&u)[3]) {
                double w[3];
                applyA0(v, w);
                UNROLL for (int k = 0; k < 3; ++k) u[k] = keep_if(w[k] + bpk[k], rmask);
            };
            double xa[3], xb[3], y[3], ra[3], rb[3], ry[3];
            UNROLL for (int k = 0; k < 3; ++k) { xa[k] = kvalid ? Xs[co + k] : 0.0; y[k] = xa[k]; }
            applyA(y, ry);
            UNROLL for (int k = 0; k < 3; ++k) ra[k] = ry[k];
            // the other group's rows the gradient needs, fetched as soon as ry exists: the LDS crossbar's latency then
            // passes under the norms / reductions of the iteration before instead of in front of the gradient
            double fy[3];
            UNROLL for (int k = 0; k < 3; ++k) fy[k] = bperm(ry[k], src2);
            double invL = 2.0 * (1.0 / L_x);
            const bool banded_x = !a.exact_step_decisions && rho >= 0.0 && !__any(qd[0] < 0 || qd[1] < 0 || qd[2] < 0);
            bool act = true;
            auto iterate = [&](const double (&xo)[3], const double (&ro)[3], double (&xn)[3], double (&rn)[3], int i) {
                const double cm = cmtab[i];
                bool done;
                for (;;) {
                    {   // half gradient Q y + q/2 + rho A_f^T (A_f y + bPk), step, box projection (fista.cpp:10)
                        double z[3];
                        UNROLL for (int k = 0; k < 3; ++k) z[k] = ry[k] - from_prev(ry[k]);
                        UNROLL for (int k = 0; k < 3; ++k) z[k] += fma(a2[k], fy[k], fma(b2[k], fy[(k + 1) % 3], c2[k] * fy[(k + 2) % 3]));
                        UNROLL for (int k = 0; k < 3; ++k) {
                            const double g = fma(qd[k], y[k], fma(rho, z[k], q[k]));
                            xn[k] = clamp_box(fma(-g, invL, y[k]), lb[k], ub[k]);
                        }
                    }
                    applyA(xn, rn);
                    double g2 = 0, cv = 0, e2 = 0;
                    UNROLL for (int k = 0; k < 3; ++k) {
                        const double d = xn[k] - y[k], e = rn[k] - ry[k];
                        g2 = fma(d, d, g2);
                        cv = fma(qd[k] * d, d, cv);
                        e2 = fma(e, e, e2);
                    }
                    cv = fma(rho, e2, cv);
                    const double Lh = L_x * 0.5;
                    bool bt;
                    if (!(banded_x && banded_decisions(g2, cv, Lh, tol2, bt, done))) {
                        double g2s = g2, cvs = cv;
                        seg_sum2<64>(g2s, cvs);
                        const double rhs = Lh * g2s;
                        bt = cvs > rhs;
                        done = g2s < tol2;
                        if ((fabs(cvs - rhs) <= 1e-14 * rhs) || (fabs(g2s - tol2) <= 1e-14 * tol2)) {
                            const double Gn = sqrt(g2s);
                            bt = cvs > Lh * (Gn * Gn);
                            done = Gn < tol;
                        }
                    }
                    if (!__any(bt)) break;
                    L_x *= beta; ++bt_x;
                    invL = 2.0 * (1.0 / L_x);
                }
                done = __any(done);
                if ((done || i == maxit - 1) && kvalid) { UNROLL for (int k = 0; k < 3; ++k) Xs[co + k] = xn[k]; }
                UNROLL for (int k = 0; k < 3; ++k) ry[k] = fma(cm, rn[k] - ro[k], rn[k]);
                UNROLL for (int k = 0; k < 3; ++k) fy[k] = bperm(ry[k], src2);
                UNROLL for (int k = 0; k < 3; ++k) y[k] = fma(cm, xn[k] - xo[k], xn[k]);
                ++it_x;
                act = !done;
            };
            for (int i = 0; i < maxit && act; i += 2) {
                iterate(xa, ra, xb, rb, i);
                if (i + 1 >= maxit || !act) break;
                iterate(xb, rb, xa, ra, i + 1);
            }
            // dyn_violation = A_f X - b_f ; P += dyn_violation          (biconvex.cpp:98-99)
            double fin[3], w[3], v2 = 0;
            UNROLL for (int k = 0; k < 3; ++k) fin[k] = kvalid ? Xs[co + k] : 0.0;
            applyA0(fin, w);
            UNROLL for (int k = 0; k < 3; ++k) {
                const double d = rvalid ? (w[k] - bf[k]) : 0.0;
                const double di = x0lane ? (fin[k] - a.x_init[pb * 9 + 3 * xg + k]) : 0.0;
                if (rvalid) Ps[co + k] += d;
                if (x0lane) Ps[9 * H + 3 * xg + k] += di;
                v2 += d * d + di * di;
            }
            v2 = seg_sum<64>(v2);
            const double nrm = sqrt(v2);
            last_viol = nrm;
            ++n_admm;
            if (a.hist && lane == 0) a.hist[pb * a.c.num_iters + it] = nrm;
            if (a.trace && lane == 0) {
                int *tr = a.trace + (pb * a.c.num_iters + it) * 4;
                tr[0] = it_f; tr[1] = it_x; tr[2] = bt_f; tr[3] = bt_x;
            }
            const bool isn = __any(isnan(nrm));
            if (isn) status = 2;                                              // biconvex.cpp:106-109
            if (isn || __any(nrm < exit_tol)) alive = false;                  // biconvex.cpp:111-114
        }
        __syncthreads();
    }

    // ---- results
    for (long i = lane; i < nx; i += 64) { a.X[pb * nx + i] = Xs[i]; a.P[pb * nx + i] = Ps[i]; }
    for (long i = lane; i < nf; i += 64) a.F[pb * nf + i] = Fs[i];
    if (lane == 0) {
        a.L_x[pb] = L_x;
        a.L_f[pb] = L_f;
        if (a.dyn_viol) a.dyn_viol[pb] = last_viol;
        if (a.stats) {
            int *s = a.stats + pb * kStats;
            s[0] = n_admm; s[1] = it_f; s[2] = it_x; s[3] = bt_f; s[4] = bt_x; s[5] = status;
        }
    }
}

template <bool RAW, bool HASQF>
hipError_t launch(const BatchArgs &a, hipStream_t stream) {
    const size_t nstate = 2 * 9 * (size_t)(a.H + 1) + 12 * (size_t)a.H;
    const size_t lds = sizeof(double) * ((((size_t)a.c.maxit + 1) & ~(size_t)1) + nstate);
    hipLaunchKernelGGL((biconvex_latency_kernel<RAW, HASQF>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

bool latency_mapping_fits(const BatchArgs &a, int n_eff) {
    return n_eff == 4 && a.precision == 0 && a.H >= 1 && a.H + 1 <= kLatKnots && a.B >= 1;
}

hipError_t launch_biconvex_latency(const BatchArgs &a, hipStream_t stream) {
    if (!a.raw) return launch<false, false>(a, stream);
    return a.qf ? launch<true, true>(a, stream) : launch<true, false>(a, stream);
}

}  // namespace bunmpc

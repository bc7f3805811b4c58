// Device-side rigid-body and SE(3) routines for the IK-DDP kernels (one thread = one node).
// GPU restatement of what the reference takes from pinocchio 2.6.9 (ISL/src/ik/action_model.cpp:60-63,
// 82-86; ISL/src/motion_planner/kino_dyn.cpp:42): conventions and formulas are those of
// oracle/rbd_np.py, which pins them by finite differences (pinocchio itself is absent: parity unpinned).
#pragma once
#include "ik_types.h"

namespace bunmpc {
namespace rbd {

#define RBD_D __device__ __forceinline__
#define UNROLL_RBD_DEV _Pragma("unroll")

RBD_D void cross3(const double *a, const double *b, double *c) {
    const double c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
    c[0] = c0; c[1] = c1; c[2] = c2;
}
RBD_D double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// C = A B (3x3 row-major)
RBD_D void mat3mul(const double *A, const double *B, double *C) {
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; ++i) C[i] = t[i];
}
RBD_D void mat3vec(const double *A, const double *v, double *o) {
    const double o0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], o1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2],
                 o2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = o0; o[1] = o1; o[2] = o2;
}
RBD_D void mat3Tvec(const double *A, const double *v, double *o) {
    const double o0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2], o1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2],
                 o2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = o0; o[1] = o1; o[2] = o2;
}
RBD_D void skew3(const double *v, double *K) {
    K[0] = 0; K[1] = -v[2]; K[2] = v[1]; K[3] = v[2]; K[4] = 0; K[5] = -v[0]; K[6] = -v[1]; K[7] = v[0]; K[8] = 0;
}

// sin and cos together, ~1 ulp, for the moderate arguments this code meets (joint angles, rotation angles <= pi;
// fine up to |x| ~ 1e5): Cody-Waite reduction by pi/2 in two pieces + the fdlibm kernel polynomials.  The library
// sin / cos carry a Payne-Hanek path for huge arguments and do not share their range reduction.
RBD_D void sincos_fast(double x, double &s, double &c) {
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double sr = fma(r * z, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int n = (int)k & 3;
    const double ss = (n & 1) ? cr : sr, cc = (n & 1) ? sr : cr;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

// coefficients a = sin t / t, b = (1 - cos t)/t^2, c = (t - sin t)/t^3
RBD_D void abc(double t2, double &a, double &b, double &c) {
    if (t2 < 1e-6) {
        a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
        b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        c = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        const double t = sqrt(t2);
        double st, ct;
        sincos_fast(t, st, ct);
        a = st / t; b = (1.0 - ct) / t2; c = (t - st) / (t2 * t);
    }
}
RBD_D void exp3(const double *w, double *R) {
    double a, b, c, K[9], K2[9];
    abc(dot3(w, w), a, b, c);
    skew3(w, K); mat3mul(K, K, K2);
    for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
    R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}
// exp6(nu = (v, w)) -> R, p = V(w) v
RBD_D void exp6(const double *nu, double *R, double *p) {
    double a, b, c, K[9], K2[9], V[9];
    abc(dot3(nu + 3, nu + 3), a, b, c);
    skew3(nu + 3, K); mat3mul(K, K, K2);
    for (int i = 0; i < 9; ++i) { R[i] = a * K[i] + b * K2[i]; V[i] = b * K[i] + c * K2[i]; }
    R[0] += 1.0; R[4] += 1.0; R[8] += 1.0; V[0] += 1.0; V[4] += 1.0; V[8] += 1.0;
    mat3vec(V, nu, p);
}
RBD_D void log3(const double *R, double *w) {
    const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double nv = sqrt(dot3(v, v));
    const double t = atan2(0.5 * nv, 0.5 * (R[0] + R[4] + R[8] - 1.0));
    double f;
    if (t < 1e-3) f = 0.5 * (1.0 + t * t / 6.0 + 7.0 * t * t * t * t / 360.0);
    else if (3.141592653589793 - t < 1e-6) {  // near pi: axis from the symmetric part
        const double A[3] = {0.5 * (R[0] + 1.0), 0.5 * (R[4] + 1.0), 0.5 * (R[8] + 1.0)};
        int k = A[0] >= A[1] ? (A[0] >= A[2] ? 0 : 2) : (A[1] >= A[2] ? 1 : 2);
        double ax[3] = {0.25 * (R[k] + R[3 * k]), 0.25 * (R[3 + k] + R[3 * k + 1]), 0.25 * (R[6 + k] + R[3 * k + 2])};
        ax[k] = A[k];
        const double s = 1.0 / sqrt(A[k]);
        double sg = (ax[0] * v[0] + ax[1] * v[1] + ax[2] * v[2]) < 0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) w[i] = sg * t * ax[i] * s;
        return;
    } else { double st, ct; sincos_fast(t, st, ct); f = t / (2.0 * st); }
    w[0] = f * v[0]; w[1] = f * v[1]; w[2] = f * v[2];
}
RBD_D double beta_of(double t2) {
    if (t2 < 1e-6) return 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
    const double t = sqrt(t2);
    double st, ct;
    sincos_fast(t, st, ct);
    return 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
}
RBD_D void log6(const double *R, const double *p, double *nu) {
    double K[9], K2[9];
    log3(R, nu + 3);
    const double beta = beta_of(dot3(nu + 3, nu + 3));
    skew3(nu + 3, K); mat3mul(K, K, K2);
    double Vi[9];
    for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * K[i] + beta * K2[i];
    Vi[0] += 1.0; Vi[4] += 1.0; Vi[8] += 1.0;
    mat3vec(Vi, p, nu);
}
RBD_D void jlog3(const double *w, double *J) {
    const double t2 = dot3(w, w);
    double alpha, diag;
    if (t2 < 1e-6) { alpha = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0; diag = 0.5 * (2.0 - t2 / 6.0 - t2 * t2 / 360.0); }
    else { const double t = sqrt(t2); double st, ct; sincos_fast(t, st, ct); const double s1c = st / (1.0 - ct); alpha = 1.0 / t2 - s1c / (2.0 * t); diag = 0.5 * t * s1c; }
    double K[9]; skew3(w, K);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[3 * i + j] = alpha * w[i] * w[j] + 0.5 * K[3 * i + j];
    J[0] += diag; J[4] += diag; J[8] += diag;
}
RBD_D void jexp3(const double *w, double *J) {
    double a, b, c, K[9], K2[9];
    abc(dot3(w, w), a, b, c);
    skew3(w, K); mat3mul(K, K, K2);
    for (int i = 0; i < 9; ++i) J[i] = -b * K[i] + c * K2[i];
    J[0] += 1.0; J[4] += 1.0; J[8] += 1.0;
}
// Barfoot's Q block of the left SE(3) Jacobian, xi = (rho, phi)
RBD_D void q_left(const double *rho, const double *phi, double *Q) {
    const double t2 = dot3(phi, phi);
    double c1, c2, c3;
    if (t2 < 1e-4) {
        c1 = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
        c2 = 1.0 / 24.0 - t2 / 720.0 + t2 * t2 / 40320.0;
        c3 = 1.0 / 120.0 - t2 / 2520.0 + t2 * t2 / 120960.0;
    } else {
        const double t = sqrt(t2);
        double st, ct;
        sincos_fast(t, st, ct);
        c1 = (t - st) / (t2 * t);
        c2 = (t2 + 2.0 * ct - 2.0) / (2.0 * t2 * t2);
        c3 = (2.0 * t - 3.0 * st + t * ct) / (2.0 * t2 * t2 * t);
    }
    double P[9], Rh[9], PR[9], RP[9], PRP[9], PPR[9], RPP[9], PRPP[9], PPRP[9];
    skew3(phi, P); skew3(rho, Rh);
    mat3mul(P, Rh, PR); mat3mul(Rh, P, RP); mat3mul(PR, P, PRP);
    mat3mul(P, PR, PPR); mat3mul(RP, P, RPP); mat3mul(PRP, P, PRPP); mat3mul(P, PRP, PPRP);
    for (int i = 0; i < 9; ++i)
        Q[i] = 0.5 * Rh[i] + c1 * (PR[i] + RP[i] + PRP[i]) + c2 * (PPR[i] + RPP[i] - 3.0 * PRP[i]) + c3 * (PRPP[i] + PPRP[i]);
}
// right Jacobian of exp6 (6x6 row-major): [[Jr, Q(-nu)],[0, Jr]]
RBD_D void jexp6(const double *nu, double *J) {
    double Jr[9], Q[9];
    const double mn[6] = {-nu[0], -nu[1], -nu[2], -nu[3], -nu[4], -nu[5]};
    jexp3(nu + 3, Jr); q_left(mn, mn + 3, Q);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        J[6 * i + j] = Jr[3 * i + j]; J[6 * (i + 3) + 3 + j] = Jr[3 * i + j];
        J[6 * i + 3 + j] = Q[3 * i + j]; J[6 * (i + 3) + j] = 0.0;
    }
}
// d log6(M exp6(d))/dd at 0 = [[A, -A Q A],[0, A]], A = jlog3(w), Q = q_left(-nu)
RBD_D void jlog6_of(const double *nu, double *J) {   // nu = log6(M), already at hand
    double A[9], Q[9], AQ[9], AQA[9];
    const double mn[6] = {-nu[0], -nu[1], -nu[2], -nu[3], -nu[4], -nu[5]};
    jlog3(nu + 3, A); q_left(mn, mn + 3, Q);
    mat3mul(A, Q, AQ); mat3mul(AQ, A, AQA);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        J[6 * i + j] = A[3 * i + j]; J[6 * (i + 3) + 3 + j] = A[3 * i + j];
        J[6 * i + 3 + j] = -AQA[3 * i + j]; J[6 * (i + 3) + j] = 0.0;
    }
}
// 6x6 action of M^-1 on motions: [[R^T, -R^T [p]x],[0, R^T]]
RBD_D void act_inv(const double *R, const double *p, double *X) {
    double K[9]; skew3(p, K);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        const double rt = R[3 * j + i];
        X[6 * i + j] = rt; X[6 * (i + 3) + 3 + j] = rt; X[6 * (i + 3) + j] = 0.0;
        X[6 * i + 3 + j] = -(R[i] * K[j] + R[3 + i] * K[3 + j] + R[6 + i] * K[6 + j]);
    }
}
RBD_D void quat_to_R(const double *qin, double *R) {
    const double n = 1.0 / sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
    const double x = qin[0] * n, y = qin[1] * n, z = qin[2] * n, w = qin[3] * n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
// x (+) dx on the 37-dim state (free-flyer: M exp6; quaternion by right product with exp(w))
RBD_D void state_integrate(const double *x, const double *dx, double *xn) {
    double Rb[9], dR[9], dp[3], t[3];
    quat_to_R(x + 3, Rb);
    exp6(dx, dR, dp);
    mat3vec(Rb, dp, t);
    xn[0] = x[0] + t[0]; xn[1] = x[1] + t[1]; xn[2] = x[2] + t[2];
    // quaternion of exp3(w): (sin(t/2)/t w, cos(t/2))
    const double t2 = dot3(dx + 3, dx + 3), th = sqrt(t2);
    double s, cw;
    if (t2 < 1e-8) { s = 0.5 - t2 / 48.0; cw = 1.0 - t2 / 8.0; } else { double sh; sincos_fast(0.5 * th, sh, cw); s = sh / th; }
    const double d[4] = {s * dx[3], s * dx[4], s * dx[5], cw};
    const double nq = 1.0 / sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
    const double q[4] = {x[3] * nq, x[4] * nq, x[5] * nq, x[6] * nq};
    double r[4] = {q[3] * d[0] + q[0] * d[3] + q[1] * d[2] - q[2] * d[1],
                   q[3] * d[1] - q[0] * d[2] + q[1] * d[3] + q[2] * d[0],
                   q[3] * d[2] + q[0] * d[1] - q[1] * d[0] + q[2] * d[3],
                   q[3] * d[3] - q[0] * d[0] - q[1] * d[1] - q[2] * d[2]};
    const double nr = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    for (int i = 0; i < 4; ++i) xn[3 + i] = r[i] * nr;
    _Pragma("unroll") for (int i = 0; i < kNV - 6; ++i) xn[7 + i] = x[7 + i] + dx[6 + i];
    _Pragma("unroll") for (int i = 0; i < kNV; ++i) xn[kNQ + i] = x[kNQ + i] + dx[kNV + i];
}
// d = x1 (-) x0 (36); optionally the 6x6 Jlog6 of the base block (d diff / d x1)
RBD_D void state_diff_q(const double *x0, const double *x1, double *d);
template <bool JAC>
RBD_D void state_diff(const double *x0, const double *x1, double *d, double *Jl) {
    if (JAC) {      // the difference by the quaternion route (below), its Jacobian block from the difference
        state_diff_q(x0, x1, d);
        jlog6_of(d, Jl);
        return;
    }
    double R0[9], R1[9], Rr[9], dp[3], pr[3];
    quat_to_R(x0 + 3, R0); quat_to_R(x1 + 3, R1);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        Rr[3 * i + j] = R0[i] * R1[j] + R0[3 + i] * R1[3 + j] + R0[6 + i] * R1[6 + j];
    dp[0] = x1[0] - x0[0]; dp[1] = x1[1] - x0[1]; dp[2] = x1[2] - x0[2];
    mat3Tvec(R0, dp, pr);
    log6(Rr, pr, d);
    _Pragma("unroll") for (int i = 0; i < kNV - 6; ++i) d[6 + i] = x1[7 + i] - x0[7 + i];
    _Pragma("unroll") for (int i = 0; i < kNV; ++i) d[kNV + i] = x1[kNQ + i] - x0[kNQ + i];
    if (JAC) jlog6_of(d, Jl);
}

// ---- the same two state operators without rotation matrices, for the serial chains of the forward pass (x -> dx -> u -> x+):
// quaternion algebra, reciprocals by v_rcp / v_rsq + Newton steps instead of the IEEE division / sqrt sequences (~25
// dependent instructions each), one arctangent and no sine / cosine in the difference, one sincos of the half angle in the
// step.  Same mathematics as state_diff<false> / state_integrate above (equal to a few ulp: tests/test_rbd_gpu.py), a third of
// their instructions.
RBD_D double rcp_fast(double b) {       // 1 / b to ~1 ulp
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    return fma(fma(-b, r, 1.0), r, r);
}
RBD_D double rsqrt_fast(double a) {     // 1 / sqrt(a) to ~1 ulp, a > 0 and far from the subnormals
    double r = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    r = fma(fma(-h * r, r, 0.5), r, r);
    return fma(fma(-h * r, r, 0.5), r, r);
}
// atan(y / x) for x, y >= 0 (not both 0): in [0, pi/2].  fdlibm's atan polynomial on a reduced argument.
RBD_D double atan2_pos(double y, double x) {
    const bool swap = y > x;
    const double num = swap ? x : y, den = swap ? y : x;      // t = num / den in [0, 1]
    double t = num * rcp_fast(den);
    // reduce [0, 1] to |t| <= tan(pi/8): atan(t) = pi/4 + atan((t - 1)/(t + 1)) for t > tan(pi/8)
    const bool hi = t > 0.4142135623730950488;
    const double tr = hi ? (t - 1.0) * rcp_fast(t + 1.0) : t;
    const double z = tr * tr, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    double a = tr - tr * (s1 + s2);
    a = hi ? 7.85398163397448278999e-01 + a : a;
    return swap ? 1.57079632679489655800e+00 - a : a;
}
// rotate d by the unit quaternion (v, w): R d = d + 2 w (v x d) + 2 v x (v x d); transposed: the same with -v
RBD_D void quat_rotate(const double *v, double w, const double *d, double *o) {
    double t[3], u[3];
    cross3(v, d, t);
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) t[i] *= 2.0;
    cross3(v, t, u);
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) o[i] = d[i] + w * t[i] + u[i];
}
RBD_D void state_diff_q(const double *x0, const double *x1, double *d) {
    // relative placement M0^-1 M1 as a unit quaternion and a translation in frame 0
    const double n0 = rsqrt_fast(x0[3] * x0[3] + x0[4] * x0[4] + x0[5] * x0[5] + x0[6] * x0[6]);
    const double n1 = rsqrt_fast(x1[3] * x1[3] + x1[4] * x1[4] + x1[5] * x1[5] + x1[6] * x1[6]);
    const double a[3] = {x0[3] * n0, x0[4] * n0, x0[5] * n0}, aw = x0[6] * n0;
    const double b[3] = {x1[3] * n1, x1[4] * n1, x1[5] * n1}, bw = x1[6] * n1;
    double axb[3];
    cross3(a, b, axb);
    double qw = aw * bw + dot3(a, b);
    double qv[3];
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) qv[i] = aw * b[i] - bw * a[i] - axb[i];
    if (qw < 0.0) { qw = -qw; UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) qv[i] = -qv[i]; }      // the rotation by at most pi
    const double n2 = dot3(qv, qv);
    double w[3], beta;
    if (n2 < 1e-16) {      // theta = 2 asin(n) ~ 2 n: w = 2 v; beta -> 1/12
        UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) w[i] = 2.0 * qv[i];
        beta = 1.0 / 12.0;
    } else {
        const double in = rsqrt_fast(n2), n = n2 * in;
        const double th = 2.0 * atan2_pos(n, qw), ith = rcp_fast(th);
        const double f = th * in;
        UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) w[i] = f * qv[i];
        // beta = 1/theta^2 - cot(theta/2)/(2 theta), cot(theta/2) = qw / n; series where the two terms cancel
        beta = th * th < 1e-6 ? 1.0 / 12.0 + th * th / 720.0 : ith * ith - 0.5 * (qw * in) * ith;
    }
    const double na[3] = {-a[0], -a[1], -a[2]};
    const double dp[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    double pr[3], wp[3], wwp[3];
    quat_rotate(na, aw, dp, pr);              // R0^T (p1 - p0)
    cross3(w, pr, wp);
    cross3(w, wp, wwp);
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) { d[i] = pr[i] - 0.5 * wp[i] + beta * wwp[i]; d[3 + i] = w[i]; }
    UNROLL_RBD_DEV for (int i = 0; i < kNV - 6; ++i) d[6 + i] = x1[7 + i] - x0[7 + i];
    UNROLL_RBD_DEV for (int i = 0; i < kNV; ++i) d[kNV + i] = x1[kNQ + i] - x0[kNQ + i];
}
RBD_D void state_integrate_q(const double *x, const double *dx, double *xn) {
    const double *v = dx, *w = dx + 3;
    const double t2 = dot3(w, w);
    double sh_t, ch, bb, cc;          // sin(theta/2)/theta, cos(theta/2), (1 - cos theta)/theta^2, (theta - sin theta)/theta^3
    if (t2 < 1e-6) {
        sh_t = 0.5 - t2 / 48.0 + t2 * t2 / 3840.0; ch = 1.0 - t2 / 8.0 + t2 * t2 / 384.0;
        bb = 0.5 - t2 / 24.0 + t2 * t2 / 720.0; cc = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        const double it = rsqrt_fast(t2), th = t2 * it;
        double sh;
        sincos_fast(0.5 * th, sh, ch);
        sh_t = sh * it;
        bb = 2.0 * sh_t * sh_t;                              // 2 sin^2(theta/2) / theta^2
        cc = (th - 2.0 * sh * ch) * it * it * it;            // (theta - sin theta) / theta^3
    }
    const double nq = rsqrt_fast(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
    const double q[3] = {x[3] * nq, x[4] * nq, x[5] * nq}, qw = x[6] * nq;
    // p+ = p + R(q) (v + b w x v + c w x (w x v))
    double wv[3], wwv[3], u[3], ru[3];
    cross3(w, v, wv);
    cross3(w, wv, wwv);
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) u[i] = v[i] + bb * wv[i] + cc * wwv[i];
    quat_rotate(q, qw, u, ru);
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) xn[i] = x[i] + ru[i];
    // q+ = q (x) (sin(theta/2)/theta w, cos(theta/2)), normalised
    const double dv[3] = {sh_t * w[0], sh_t * w[1], sh_t * w[2]};
    double qxd[3];
    cross3(q, dv, qxd);
    double r[4];
    UNROLL_RBD_DEV for (int i = 0; i < 3; ++i) r[i] = qw * dv[i] + ch * q[i] + qxd[i];
    r[3] = qw * ch - dot3(q, dv);
    const double nr = rsqrt_fast(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    UNROLL_RBD_DEV for (int i = 0; i < 4; ++i) xn[3 + i] = r[i] * nr;
    UNROLL_RBD_DEV for (int i = 0; i < kNV - 6; ++i) xn[7 + i] = x[7 + i] + dx[6 + i];
    UNROLL_RBD_DEV for (int i = 0; i < kNV; ++i) xn[kNQ + i] = x[kNQ + i] + dx[kNV + i];
}

// composite inertia about the world origin: mass, first moment h1 = m c, I_O (xx xy xz yy yz zz)
struct Comp { double m, h1[3], I[6]; };
RBD_D void comp_zero(Comp &c) { c.m = 0; for (int i = 0; i < 3; ++i) c.h1[i] = 0; for (int i = 0; i < 6; ++i) c.I[i] = 0; }
RBD_D void comp_add(Comp &c, const Comp &o) { c.m += o.m; for (int i = 0; i < 3; ++i) c.h1[i] += o.h1[i]; for (int i = 0; i < 6; ++i) c.I[i] += o.I[i]; }
// (f, n_O) = I * (v, w)
RBD_D void comp_apply(const Comp &c, const double *mot, double *h) {
    const double *v = mot, *w = mot + 3;
    double t[3];
    cross3(w, c.h1, t);
    h[0] = c.m * v[0] + t[0]; h[1] = c.m * v[1] + t[1]; h[2] = c.m * v[2] + t[2];
    cross3(c.h1, v, t);
    h[3] = c.I[0] * w[0] + c.I[1] * w[1] + c.I[2] * w[2] + t[0];
    h[4] = c.I[1] * w[0] + c.I[3] * w[1] + c.I[4] * w[2] + t[1];
    h[5] = c.I[2] * w[0] + c.I[4] * w[1] + c.I[5] * w[2] + t[2];
}

// Kinematic quantities of one state.  Bodies: 0 = base, i+1 = joint i.
struct Kin {
    double oR[kMaxJoints + 1][9], op[kMaxJoints + 1][3];
    double S[kNV][6];                 // columns (v_O, w) in world coordinates
    double V[kMaxJoints + 1][6];      // body twists
    Comp body[kMaxJoints + 1];
    double hb[kMaxJoints + 1][6];     // body momenta about the world origin
    double M, com[3], hO[6], hg[6];
};

template <bool VEL>
RBD_D void kin_compute(const RobotModelDev &m, const double *x, Kin &k) {
    quat_to_R(x + 3, k.oR[0]);
    k.op[0][0] = x[0]; k.op[0][1] = x[1]; k.op[0][2] = x[2];
    for (int a = 0; a < 3; ++a) {
        const double e[3] = {k.oR[0][a], k.oR[0][3 + a], k.oR[0][6 + a]};
        for (int i = 0; i < 3; ++i) { k.S[a][i] = e[i]; k.S[a][3 + i] = 0.0; k.S[3 + a][3 + i] = e[i]; }
        cross3(k.op[0], e, k.S[3 + a]);
    }
    for (int i = 0; i < m.nj; ++i) {
        const int b = m.parent[i] + 1;
        double Rq[9], Rl[9], aq[3] = {m.axis[i][0] * x[7 + i], m.axis[i][1] * x[7 + i], m.axis[i][2] * x[7 + i]}, t[3];
        exp3(aq, Rq);
        mat3mul(m.R[i], Rq, Rl);
        mat3mul(k.oR[b], Rl, k.oR[i + 1]);
        mat3vec(k.oR[b], m.p[i], t);
        for (int c = 0; c < 3; ++c) k.op[i + 1][c] = t[c] + k.op[b][c];
        mat3vec(k.oR[i + 1], m.axis[i], k.S[6 + i] + 3);
        cross3(k.op[i + 1], k.S[6 + i] + 3, k.S[6 + i]);
    }
    const double *v = x + kNQ;
    if (VEL) {
        for (int c = 0; c < 6; ++c) { double s = 0; for (int a = 0; a < 6; ++a) s += k.S[a][c] * v[a]; k.V[0][c] = s; }
        for (int i = 0; i < m.nj; ++i)
            for (int c = 0; c < 6; ++c) k.V[i + 1][c] = k.V[m.parent[i] + 1][c] + k.S[6 + i][c] * v[6 + i];
    }
    double h1[3] = {0, 0, 0};
    k.M = 0;
    for (int c = 0; c < 6; ++c) k.hO[c] = 0;
    for (int b = 0; b <= m.nj; ++b) {
        double cw[3], RI[9], Iw[9];
        mat3vec(k.oR[b], m.com[b], cw);
        for (int c = 0; c < 3; ++c) cw[c] += k.op[b][c];
        const double I[9] = {m.inertia[b][0], m.inertia[b][1], m.inertia[b][2], m.inertia[b][1], m.inertia[b][3], m.inertia[b][4],
                             m.inertia[b][2], m.inertia[b][4], m.inertia[b][5]};
        mat3mul(k.oR[b], I, RI);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            Iw[3 * i + j] = RI[3 * i] * k.oR[b][3 * j] + RI[3 * i + 1] * k.oR[b][3 * j + 1] + RI[3 * i + 2] * k.oR[b][3 * j + 2];
        const double mb = m.mass[b], cc = dot3(cw, cw);
        Comp &cb = k.body[b];
        cb.m = mb;
        for (int c = 0; c < 3; ++c) cb.h1[c] = mb * cw[c];
        cb.I[0] = Iw[0] + mb * (cc - cw[0] * cw[0]); cb.I[1] = Iw[1] - mb * cw[0] * cw[1]; cb.I[2] = Iw[2] - mb * cw[0] * cw[2];
        cb.I[3] = Iw[4] + mb * (cc - cw[1] * cw[1]); cb.I[4] = Iw[5] - mb * cw[1] * cw[2]; cb.I[5] = Iw[8] + mb * (cc - cw[2] * cw[2]);
        k.M += mb;
        for (int c = 0; c < 3; ++c) h1[c] += cb.h1[c];
        if (VEL) { comp_apply(cb, k.V[b], k.hb[b]); for (int c = 0; c < 6; ++c) k.hO[c] += k.hb[b][c]; }
    }
    for (int c = 0; c < 3; ++c) k.com[c] = h1[c] / k.M;
    if (VEL) {
        double t[3];
        cross3(k.com, k.hO, t);
        for (int c = 0; c < 3; ++c) { k.hg[c] = k.hO[c]; k.hg[3 + c] = k.hO[3 + c] - t[c]; }
    }
}

RBD_D void frame_position(const RobotModelDev &m, const Kin &k, int f, double *x) {
    const int b = m.frame_body[f];
    mat3vec(k.oR[b], m.frame_p[f], x);
    for (int c = 0; c < 3; ++c) x[c] += k.op[b][c];
}
// does velocity column col move body b (serial chains off the base)?
RBD_D bool in_support(const RobotModelDev &m, int b, int col) {
    if (col < 6) return true;
    if (b == 0) return false;
    int j = b - 1;
    while (j >= 0) { if (j == col - 6) return true; j = m.parent[j]; }
    return false;
}

}  // namespace rbd
}  // namespace bunmpc

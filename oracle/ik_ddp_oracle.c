/*
 * oracle/ik_ddp_oracle.c -- compiled CPU restatement of the reference's whole-body inverse
 * kinematics: ik::InverseKinematics::optimize and what it delegates to pinocchio / crocoddyl.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the
 * product path never includes, links or calls this file.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in crocoddyl 1.9.0 and pinocchio 2.6.9
 * (.devcontainer/Dockerfile:84-92), neither of which is under /root/reference nor installable in
 * this image, and the reference holds no vectors for it.  This file is the SECOND, independently
 * written restatement (the first is oracle/ik_ddp_np.py + oracle/rbd_np.py); tests/test_ik_twin_cpu.py
 * requires the two to take the same discrete path (DDP iterations, step lengths, regularisation
 * sequence) and to agree to ~1e-10 on states / controls / cost.  Written from the mathematics, with
 * different data structures on purpose: 6x6 spatial inertias about the world origin and composite
 * sums where the numpy twin carries (mass, centre, rotational inertia) triples; the SE(3) Jacobians
 * from the ad-series where the numpy twin uses Barfoot's closed forms; dense crocoddyl-shaped
 * F_x / F_u products in the Riccati recursion.
 *
 * Reference call sites followed (paths under iterative_supervised_learning/):
 *   src/ik/inverse_kinematics.cpp:37-52   one IntegratedActionModelEuler(dt_i) per running node + terminal model
 *   src/ik/inverse_kinematics.cpp:54-71   ShootingProblem(x0, ...), SolverDDP(problem).solve() with all defaults
 *   src/ik/action_model.cpp:43-94         differential model: xout = u, Fx = 0, Fu = I, cost sum
 *   src/ik/end_effector_tasks.cpp:8-37    ResidualModelFrameTranslation
 *   src/ik/com_tasks.cpp:8-50             ResidualModelCoMPosition, ResidualModelCentroidalMomentum
 *   src/ik/regularization_costs.cpp:8-93  ResidualModelState + ActivationModelWeightedQuad, ResidualModelControl (u_ref ignored)
 *   src/motion_planner/kino_dyn.cpp:42    computeCentroidalMomentum(q, v) -> [com, vcom, hg.angular]
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXJ 12
#define MAXB (MAXJ + 1)
#define MAXV (MAXJ + 6)
#define MAXDX (2 * MAXV)
#define MAXFR 64
#define NTASK 33        /* doubles per node task block: 4 x {w, frame, ref3} | com {w, ref3} | mom {w, ref6} | w_state | w_ctrl */
#define NSLOT 4

typedef struct {
    int nj, nframes;
    int parent[MAXJ];
    double R[MAXJ][9], p[MAXJ][3], axis[MAXJ][3];
    double mass[MAXB], com[MAXB][3], inertia[MAXB][9];
    int frame_body[MAXFR];
    double frame_p[MAXFR][3];
} ikor_model_t;

/* ------------------------------------------------------------------ small vectors --- */
static inline void cross(const double *a, const double *b, double *o) {
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void mat3_mul(const double *A, const double *B, double *C) {   /* C = A B, row-major, C may not alias */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static inline void mat3_vec(const double *A, const double *x, double *y) {
    const double a = A[0] * x[0] + A[1] * x[1] + A[2] * x[2], b = A[3] * x[0] + A[4] * x[1] + A[5] * x[2],
                 c = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
    y[0] = a; y[1] = b; y[2] = c;
}
static inline void mat3T_vec(const double *A, const double *x, double *y) {
    const double a = A[0] * x[0] + A[3] * x[1] + A[6] * x[2], b = A[1] * x[0] + A[4] * x[1] + A[7] * x[2],
                 c = A[2] * x[0] + A[5] * x[1] + A[8] * x[2];
    y[0] = a; y[1] = b; y[2] = c;
}
static inline void skew3(const double *v, double *K) {
    K[0] = 0; K[1] = -v[2]; K[2] = v[1]; K[3] = v[2]; K[4] = 0; K[5] = -v[0]; K[6] = -v[1]; K[7] = v[0]; K[8] = 0;
}

/* ------------------------------------------------------------------ SO(3) / SE(3) --- */
/* Rodrigues: R = I + a K + b K^2, K = [w]x */
static void so3_exp(const double *w, double *R) {
    const double t2 = dot3(w, w), t = sqrt(t2);
    double a, b;
    if (t < 1e-4) { a = 1.0 - t2 / 6.0; b = 0.5 - t2 / 24.0; }
    else { a = sin(t) / t; const double sh = sin(0.5 * t); b = 2.0 * sh * sh / t2; }
    double K[9], K2[9];
    skew3(w, K);
    mat3_mul(K, K, K2);
    for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
    R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}
static void so3_log(const double *R, double *w) {
    const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double s = 0.5 * sqrt(dot3(v, v)), c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    const double t = atan2(s, c);
    if (t < 1e-4) {
        const double f = 0.5 * (1.0 + t * t / 6.0);
        w[0] = f * v[0]; w[1] = f * v[1]; w[2] = f * v[2];
    } else if (M_PI - t < 1e-6) {     /* rotation by ~pi: the axis from the symmetric part */
        double A[9];
        for (int i = 0; i < 9; ++i) A[i] = 0.5 * R[i];
        A[0] += 0.5; A[4] += 0.5; A[8] += 0.5;
        int k = 0;
        if (A[4] > A[0]) k = 1;
        if (A[8] > A[4 * k]) k = 2;
        const double n = sqrt(A[4 * k]);
        double ax[3] = {A[k] / n, A[3 + k] / n, A[6 + k] / n};
        if (dot3(ax, v) < 0) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; }
        w[0] = t * ax[0]; w[1] = t * ax[1]; w[2] = t * ax[2];
    } else {
        const double f = t / (2.0 * s);
        w[0] = f * v[0]; w[1] = f * v[1]; w[2] = f * v[2];
    }
}
/* exp6(nu = (v, w)) = (R, p),  p = (I + b K + c K^2) v */
static void se3_exp(const double *nu, double *R, double *p) {
    const double *v = nu, *w = nu + 3;
    const double t2 = dot3(w, w), t = sqrt(t2);
    double b, c;
    if (t < 1e-4) { b = 0.5 - t2 / 24.0; c = 1.0 / 6.0 - t2 / 120.0; }
    else { const double sh = sin(0.5 * t); b = 2.0 * sh * sh / t2; c = (t - sin(t)) / (t2 * t); }
    double wv[3], wwv[3];
    cross(w, v, wv);
    cross(w, wv, wwv);
    for (int i = 0; i < 3; ++i) p[i] = v[i] + b * wv[i] + c * wwv[i];
    so3_exp(w, R);
}
/* log6(R, p) = (V^-1 p, w),  V^-1 = I - K/2 + beta K^2 */
static void se3_log(const double *R, const double *p, double *nu) {
    double w[3];
    so3_log(R, w);
    const double t2 = dot3(w, w), t = sqrt(t2);
    double beta;
    if (t < 1e-4) beta = 1.0 / 12.0 + t2 / 720.0;
    else beta = 1.0 / t2 - cos(0.5 * t) / (2.0 * t * sin(0.5 * t));     /* sin t / (1 - cos t) = cot(t/2) */
    double wp[3], wwp[3];
    cross(w, p, wp);
    cross(w, wp, wwp);
    for (int i = 0; i < 3; ++i) { nu[i] = p[i] - 0.5 * wp[i] + beta * wwp[i]; nu[3 + i] = w[i]; }
}
static void mat6_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += A[6 * i + k] * B[6 * k + j];
            C[6 * i + j] = s;
        }
}
/* Right Jacobian of exp6 from its series,  J_r(xi) = sum_n (-1)^n ad(xi)^n / (n+1)!,  ad(v, w) = [[ [w]x, [v]x ], [0, [w]x]]
 * ((lin, ang) ordering): exp6(xi + d) = exp6(xi) exp6(J_r d).  The increments met here are one Euler step or the offset
 * from a regularisation posture (|xi| < 1), where the series settles in under twenty terms. */
static void se3_jexp(const double *nu, double *J) {
    double ad[36], term[36], tmp[36];
    memset(ad, 0, sizeof ad);
    double Kw[9], Kv[9];
    skew3(nu + 3, Kw);
    skew3(nu, Kv);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { ad[6 * i + j] = Kw[3 * i + j]; ad[6 * (3 + i) + 3 + j] = Kw[3 * i + j]; ad[6 * i + 3 + j] = Kv[3 * i + j]; }
    memset(J, 0, 36 * sizeof(double));
    memset(term, 0, sizeof term);
    for (int i = 0; i < 6; ++i) { J[7 * i] = 1.0; term[7 * i] = 1.0; }
    for (int n = 1; n < 60; ++n) {
        mat6_mul(term, ad, tmp);
        double big = 0.0;
        for (int i = 0; i < 36; ++i) { term[i] = -tmp[i] / (double)(n + 1); J[i] += term[i]; big = fmax(big, fabs(term[i])); }
        if (big < 1e-19) break;
    }
}
/* 6x6 inverse by Gauss-Jordan with partial pivoting (well conditioned: J_r is close to the identity here) */
static void mat6_inv(const double *A, double *Ai) {
    double M[6][12];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { M[i][j] = A[6 * i + j]; M[i][6 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 6; ++c) {
        int pv = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > fabs(M[pv][c])) pv = r;
        if (pv != c) for (int j = 0; j < 12; ++j) { const double t = M[c][j]; M[c][j] = M[pv][j]; M[pv][j] = t; }
        const double d = 1.0 / M[c][c];
        for (int j = 0; j < 12; ++j) M[c][j] *= d;
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f != 0.0) for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) Ai[6 * i + j] = M[i][6 + j];
}
/* d log6(M exp6(d)) / d d at d = 0  =  J_r(log6 M)^-1 */
static void se3_jlog(const double *R, const double *p, double *J) {
    double nu[6], Jr[36];
    se3_log(R, p, nu);
    se3_jexp(nu, Jr);
    mat6_inv(Jr, J);
}
/* action of M^-1 on motions, (lin, ang) ordering: [[R^T, -R^T [p]x], [0, R^T]] */
static void se3_act_inv(const double *R, const double *p, double *X) {
    double K[9];
    skew3(p, K);
    memset(X, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double rt = R[3 * j + i];
            X[6 * i + j] = rt; X[6 * (3 + i) + 3 + j] = rt;
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += R[3 * k + i] * K[3 * k + j];
            X[6 * i + 3 + j] = -s;
        }
}
static void quat_to_R(const double *q4, double *R) {      /* xyzw, normalised first */
    const double n = sqrt(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]);
    const double x = q4[0] / n, y = q4[1] / n, z = q4[2] / n, w = q4[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void R_to_quat(const double *R, double *q) {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        const double s = sqrt(tr + 1.0) * 2.0;
        q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; q[3] = 0.25 * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double s = sqrt(1.0 + R[4 * i] - R[4 * j] - R[4 * k]) * 2.0;
        q[i] = 0.25 * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) / s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) / s;
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] /= n;
}

/* --------------------------------------------------- StateMultibody integrate / diff --- */
/* x (+) dx: free-flyer M <- M exp6(dx[0:6]) (pinocchio::integrate), joints and velocities additive */
static void state_integrate(int nj, const double *x, const double *dx, double *out) {
    const int nq = 7 + nj, nv = 6 + nj;
    double R[9], dR[9], dp[3], Rn[9], t[3];
    quat_to_R(x + 3, R);
    se3_exp(dx, dR, dp);
    mat3_vec(R, dp, t);
    for (int i = 0; i < 3; ++i) out[i] = x[i] + t[i];
    mat3_mul(R, dR, Rn);
    R_to_quat(Rn, out + 3);
    for (int i = 0; i < nj; ++i) out[7 + i] = x[7 + i] + dx[6 + i];
    for (int i = 0; i < nv; ++i) out[nq + i] = x[nq + i] + dx[nv + i];
}
/* relative placement M0^-1 M1 of the free-flyers of two states */
static void base_rel(const double *x0, const double *x1, double *R, double *p) {
    double R0[9], R1[9], d[3];
    quat_to_R(x0 + 3, R0);
    quat_to_R(x1 + 3, R1);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += R0[3 * k + i] * R1[3 * k + j];
            R[3 * i + j] = s;
        }
    for (int i = 0; i < 3; ++i) d[i] = x1[i] - x0[i];
    mat3T_vec(R0, d, p);
}
/* diff(x0, x1) = [log6(M0^-1 M1), dq joints, dv]; Jl != NULL: the 6x6 block of d diff / d x1 as well */
static void state_diff(int nj, const double *x0, const double *x1, double *d, double *Jl) {
    const int nq = 7 + nj, nv = 6 + nj;
    double R[9], p[3];
    base_rel(x0, x1, R, p);
    se3_log(R, p, d);
    for (int i = 0; i < nj; ++i) d[6 + i] = x1[7 + i] - x0[7 + i];
    for (int i = 0; i < nv; ++i) d[nv + i] = x1[nq + i] - x0[nq + i];
    if (Jl) se3_jlog(R, p, Jl);
}

/* ---------------------------------------------------------------------- kinematics --- */
typedef struct {
    double oR[MAXB][9], op[MAXB][3];   /* world placement of the joint frame of body b (0 = base) */
    double S[MAXV][6];                 /* motion subspace columns, world frame: (velocity of the point at the origin, angular) */
    double V[MAXB][6];                 /* body velocities */
    double Y[MAXB][36], Yc[MAXB][36];  /* spatial inertia about the world origin: body, composite of its subtree */
    double hb[MAXB][6], hs[MAXB][6];   /* momentum: body, subtree (force-like: linear, angular about the origin) */
    double M, com[3], hO[6], hg[6];
} kin_t;

static inline int col_body(int k) { return k < 6 ? 0 : k - 6 + 1; }   /* the body whose subtree column k moves */

static void mat6_vec(const double *A, const double *x, double *y) {
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += A[6 * i + k] * x[k];
        y[i] = s;
    }
}

static void kin_compute(const ikor_model_t *m, const double *q, const double *v, kin_t *k) {
    const int nj = m->nj, nb = nj + 1;
    quat_to_R(q + 3, k->oR[0]);
    for (int c = 0; c < 3; ++c) k->op[0][c] = q[c];
    memset(k->S, 0, sizeof k->S);
    for (int a = 0; a < 3; ++a) {
        const double e[3] = {k->oR[0][a], k->oR[0][3 + a], k->oR[0][6 + a]};   /* base axis a in the world */
        for (int c = 0; c < 3; ++c) { k->S[a][c] = e[c]; k->S[3 + a][3 + c] = e[c]; }
        cross(k->op[0], e, k->S[3 + a]);
    }
    for (int i = 0; i < nj; ++i) {
        const int b = m->parent[i] + 1;
        double Rl[9], Rq[9], w[3], t[9];
        for (int c = 0; c < 3; ++c) w[c] = m->axis[i][c] * q[7 + i];
        so3_exp(w, Rq);
        mat3_mul(m->R[i], Rq, Rl);
        mat3_mul(k->oR[b], Rl, t);
        memcpy(k->oR[i + 1], t, sizeof t);
        mat3_vec(k->oR[b], m->p[i], k->op[i + 1]);
        for (int c = 0; c < 3; ++c) k->op[i + 1][c] += k->op[b][c];
        mat3_vec(k->oR[i + 1], m->axis[i], k->S[6 + i] + 3);
        cross(k->op[i + 1], k->S[6 + i] + 3, k->S[6 + i]);
    }
    /* velocities */
    if (v) {
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            for (int a = 0; a < 6; ++a) s += k->S[a][c] * v[a];
            k->V[0][c] = s;
        }
        for (int i = 0; i < nj; ++i)
            for (int c = 0; c < 6; ++c) k->V[i + 1][c] = k->V[m->parent[i] + 1][c] + k->S[6 + i][c] * v[6 + i];
    } else memset(k->V, 0, sizeof k->V);
    /* spatial inertias about the world origin */
    k->M = 0.0;
    double mc[3] = {0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        double c[3], RI[9], Iw[9], K[9], K2[9];
        mat3_vec(k->oR[b], m->com[b], c);
        for (int a = 0; a < 3; ++a) c[a] += k->op[b][a];
        mat3_mul(k->oR[b], m->inertia[b], RI);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Iw[3 * i + j] = RI[3 * i] * k->oR[b][3 * j] + RI[3 * i + 1] * k->oR[b][3 * j + 1] + RI[3 * i + 2] * k->oR[b][3 * j + 2];
        skew3(c, K);
        mat3_mul(K, K, K2);
        const double mb = m->mass[b];
        double *Y = k->Y[b];
        memset(Y, 0, 36 * sizeof(double));
        for (int i = 0; i < 3; ++i) {
            Y[6 * i + i] = mb;
            for (int j = 0; j < 3; ++j) {
                Y[6 * i + 3 + j] = -mb * K[3 * i + j];
                Y[6 * (3 + i) + j] = mb * K[3 * i + j];
                Y[6 * (3 + i) + 3 + j] = Iw[3 * i + j] - mb * K2[3 * i + j];
            }
        }
        mat6_vec(Y, k->V[b], k->hb[b]);
        k->M += mb;
        for (int a = 0; a < 3; ++a) mc[a] += mb * c[a];
    }
    for (int a = 0; a < 3; ++a) k->com[a] = mc[a] / k->M;
    /* composites, leaves first (joints are listed parents first) */
    memcpy(k->Yc, k->Y, sizeof k->Y);
    memcpy(k->hs, k->hb, sizeof k->hb);
    for (int i = nj - 1; i >= 0; --i) {
        const int b = m->parent[i] + 1;
        for (int e = 0; e < 36; ++e) k->Yc[b][e] += k->Yc[i + 1][e];
        for (int e = 0; e < 6; ++e) k->hs[b][e] += k->hs[i + 1][e];
    }
    memcpy(k->hO, k->hs[0], sizeof k->hO);
    double t[3];
    cross(k->com, k->hO, t);
    for (int a = 0; a < 3; ++a) { k->hg[a] = k->hO[a]; k->hg[3 + a] = k->hO[3 + a] - t[a]; }
}
/* does velocity column kcol move body b? */
static int supports(const ikor_model_t *m, int kcol, int b) {
    if (kcol < 6) return 1;
    int j = b - 1;          /* joint of body b (-1 for the base) */
    while (j >= 0) { if (j == kcol - 6) return 1; j = m->parent[j]; }
    return 0;
}
static void frame_position(const ikor_model_t *m, const kin_t *k, int f, double *x) {
    const int b = m->frame_body[f];
    mat3_vec(k->oR[b], m->frame_p[f], x);
    for (int a = 0; a < 3; ++a) x[a] += k->op[b][a];
}
/* columns (3 entries each) of the world-aligned translation Jacobian of frame f; J is 3 x nv row-major */
static void frame_jacobian(const ikor_model_t *m, const kin_t *k, int f, double *J) {
    const int nv = 6 + m->nj, b = m->frame_body[f];
    double x[3];
    frame_position(m, k, f, x);
    memset(J, 0, sizeof(double) * 3 * nv);
    for (int c = 0; c < nv; ++c) {
        if (!supports(m, c, b)) continue;
        double t[3];
        cross(k->S[c] + 3, x, t);
        for (int a = 0; a < 3; ++a) J[a * nv + c] = k->S[c][a] + t[a];
    }
}
/* centroidal map A_g (6 x nv), CoM Jacobian (3 x nv) and dh_g/dq (6 x nv), row-major */
static void centroidal_derivs(const ikor_model_t *m, const kin_t *k, double *Ag, double *Jc, double *dh) {
    const int nv = 6 + m->nj;
    for (int c = 0; c < nv; ++c) {
        const int b = col_body(c);
        double h[6], t[3];
        mat6_vec(k->Yc[b], k->S[c], h);                 /* momentum of the subtree per unit rate of column c */
        cross(k->com, h, t);
        for (int a = 0; a < 3; ++a) { Ag[a * nv + c] = h[a]; Ag[(3 + a) * nv + c] = h[3 + a] - t[a]; Jc[a * nv + c] = h[a] / k->M; }
    }
    for (int c = 0; c < nv; ++c) {
        const int b = col_body(c);
        const double *S = k->S[c], *hs = k->hs[b];
        /* S x* h_sub */
        double cf[6], t1[3], t2[3];
        cross(S + 3, hs, cf);
        cross(S + 3, hs + 3, t1);
        cross(S, hs, t2);
        for (int a = 0; a < 3; ++a) cf[3 + a] = t1[a] + t2[a];
        /* S x V_parent: the velocity the subtree inherits from above the joint of column c (zero for the base columns) */
        double sv[6] = {0, 0, 0, 0, 0, 0};
        if (c >= 6) {
            const double *Vp = k->V[m->parent[c - 6] + 1];
            cross(S + 3, Vp, t1);
            cross(S, Vp + 3, t2);
            for (int a = 0; a < 3; ++a) sv[a] = t1[a] + t2[a];
            cross(S + 3, Vp + 3, sv + 3);
        }
        double Isv[6], dO[6];
        mat6_vec(k->Yc[b], sv, Isv);
        for (int a = 0; a < 6; ++a) dO[a] = cf[a] - Isv[a];
        /* re-centred at the (moving) CoM: n_g = n_O - c x f */
        const double jc[3] = {Jc[c], Jc[nv + c], Jc[2 * nv + c]};
        cross(jc, k->hO, t1);
        cross(k->com, dO, t2);
        for (int a = 0; a < 3; ++a) { dh[a * nv + c] = dO[a]; dh[(3 + a) * nv + c] = dO[3 + a] - t1[a] - t2[a]; }
    }
}

/* ---------------------------------------------------------------------- node model --- */
typedef struct {
    const ikor_model_t *m;
    int T;
    const double *dt;                  /* [T] */
    const double *tasks;               /* [T+1][NTASK] */
    const double *state_w, *x_reg, *ctrl_w;
    long sn_sw, sn_xr, sn_cw;          /* node strides (0 = one vector for all nodes) */
} ikor_problem_t;

typedef struct {      /* per-node derivative data (crocoddyl's ActionData): dense, as SolverDDP uses it */
    double cost;
    double xnext[MAXJ + 7 + MAXV];
    double Lx[MAXDX], Lu[MAXV], Lxx[MAXDX * MAXDX], Luu[MAXV * MAXV], Fx[MAXDX * MAXDX], Fu[MAXDX * MAXV];
} node_t;

/* cost of node t at (x, u) [+ derivatives when nd != NULL] and the Euler step.  Running nodes:
 * IntegratedActionModelEuler(dt) around xout = u -- dx = [v dt + u dt^2, u dt], cost and derivatives scaled by dt;
 * terminal node: calc(x), unscaled, u = 0. */
static double node_calc(const ikor_problem_t *pb, int t, const double *x, const double *u, double *xnext, node_t *nd) {
    const ikor_model_t *m = pb->m;
    const int nj = m->nj, nq = 7 + nj, nv = 6 + nj, ndx = 2 * nv;
    const int terminal = t == pb->T;
    const double *tk = pb->tasks + (long)t * NTASK;
    const double *sw = pb->state_w + pb->sn_sw * t, *xr = pb->x_reg + pb->sn_xr * t, *cw = pb->ctrl_w + pb->sn_cw * t;
    kin_t k;
    kin_compute(m, x, x + nq, &k);
    double cost = 0.0;
    double Rx[6 * MAXDX];         /* residual Jacobian rows of the cost being added */
    if (nd) {
        memset(nd->Lx, 0, sizeof(double) * ndx);
        memset(nd->Lxx, 0, sizeof(double) * ndx * ndx);
        memset(nd->Lu, 0, sizeof(double) * nv);
        memset(nd->Luu, 0, sizeof(double) * nv * nv);
    }
    double Ag[6 * MAXV], Jc[3 * MAXV], dh[6 * MAXV];
    if (nd) centroidal_derivs(m, &k, Ag, Jc, dh);
#define ADD_GN(nr, wt, r)                                                                   \
    for (int a_ = 0; a_ < (nr); ++a_) {                                                     \
        const double *row_ = Rx + a_ * ndx;                                                 \
        for (int i_ = 0; i_ < ndx; ++i_) {                                                  \
            const double wi_ = (wt) * row_[i_];                                             \
            if (wi_ == 0.0) continue;                                                       \
            nd->Lx[i_] += wi_ * (r)[a_];                                                    \
            double *L_ = nd->Lxx + (long)i_ * ndx;                                          \
            for (int j_ = 0; j_ < ndx; ++j_) L_[j_] += wi_ * row_[j_];                      \
        }                                                                                   \
    }
    for (int s = 0; s < NSLOT; ++s) {      /* ResidualModelFrameTranslation: r = oMf.translation - ref */
        const double w = tk[5 * s];
        if (w == 0.0) continue;
        const int f = (int)tk[5 * s + 1];
        double p[3], r[3];
        frame_position(m, &k, f, p);
        for (int a = 0; a < 3; ++a) r[a] = p[a] - tk[5 * s + 2 + a];
        cost += w * 0.5 * dot3(r, r);
        if (nd) {
            double J[3 * MAXV];
            frame_jacobian(m, &k, f, J);
            memset(Rx, 0, sizeof(double) * 3 * ndx);
            for (int a = 0; a < 3; ++a) memcpy(Rx + a * ndx, J + a * nv, sizeof(double) * nv);
            ADD_GN(3, w, r)
        }
    }
    {   /* ResidualModelCoMPosition */
        const double w = tk[20];
        if (w != 0.0) {
            double r[3];
            for (int a = 0; a < 3; ++a) r[a] = k.com[a] - tk[21 + a];
            cost += w * 0.5 * dot3(r, r);
            if (nd) {
                memset(Rx, 0, sizeof(double) * 3 * ndx);
                for (int a = 0; a < 3; ++a) memcpy(Rx + a * ndx, Jc + a * nv, sizeof(double) * nv);
                ADD_GN(3, w, r)
            }
        }
    }
    {   /* ResidualModelCentroidalMomentum: r = hg - ref, Rx = [dh/dq, A_g] */
        const double w = tk[24];
        if (w != 0.0) {
            double r[6], a2 = 0.0;
            for (int a = 0; a < 6; ++a) { r[a] = k.hg[a] - tk[25 + a]; a2 += r[a] * r[a]; }
            cost += w * 0.5 * a2;
            if (nd) {
                for (int a = 0; a < 6; ++a) {
                    memcpy(Rx + a * ndx, dh + a * nv, sizeof(double) * nv);
                    memcpy(Rx + a * ndx + nv, Ag + a * nv, sizeof(double) * nv);
                }
                ADD_GN(6, w, r)
            }
        }
    }
    {   /* ResidualModelState + ActivationModelWeightedQuad: r = diff(x_ref, x), Rx = Jdiff wrt the second argument */
        const double w = tk[31];
        if (w != 0.0) {
            double r[MAXDX], Jl[36], a2 = 0.0;
            state_diff(nj, xr, x, r, nd ? Jl : NULL);
            for (int i = 0; i < ndx; ++i) a2 += sw[i] * r[i] * r[i];
            cost += w * 0.5 * a2;
            if (nd) {
                /* Rx = blockdiag(Jlog6, I): Lx += w Rx^T (sw o r), Lxx += w Rx^T diag(sw) Rx */
                for (int i = 0; i < 6; ++i) {
                    double g = 0.0;
                    for (int a = 0; a < 6; ++a) g += Jl[6 * a + i] * sw[a] * r[a];
                    nd->Lx[i] += w * g;
                    for (int j = 0; j < 6; ++j) {
                        double h = 0.0;
                        for (int a = 0; a < 6; ++a) h += Jl[6 * a + i] * sw[a] * Jl[6 * a + j];
                        nd->Lxx[(long)i * ndx + j] += w * h;
                    }
                }
                for (int i = 6; i < ndx; ++i) { nd->Lx[i] += w * sw[i] * r[i]; nd->Lxx[(long)i * ndx + i] += w * sw[i]; }
            }
        }
    }
    {   /* ResidualModelControl (reference u = 0) with weighted-quadratic activation */
        const double w = tk[32];
        if (w != 0.0 && !terminal) {
            double a2 = 0.0;
            for (int i = 0; i < nv; ++i) a2 += cw[i] * u[i] * u[i];
            cost += w * 0.5 * a2;
            if (nd) for (int i = 0; i < nv; ++i) { nd->Lu[i] += w * cw[i] * u[i]; nd->Luu[(long)i * nv + i] += w * cw[i]; }
        }
    }
#undef ADD_GN
    if (terminal) { if (nd) nd->cost = cost; return cost; }
    const double dt = pb->dt[t];
    double dx[MAXDX];
    const double *v = x + nq;
    for (int i = 0; i < nv; ++i) { dx[i] = v[i] * dt + u[i] * dt * dt; dx[nv + i] = u[i] * dt; }
    state_integrate(nj, x, dx, xnext);
    if (nd) {
        /* Fx = Jint_x + Jint_dx d(dx)/dx,  Fu = Jint_dx [dt^2 I; dt I]:  Jint_x = blockdiag(Ad(exp6(dx_b))^-1, I),
         * Jint_dx = blockdiag(J_r(dx_b), I) on the free-flyer block */
        double R[9], p[3], A6[36], B6[36];
        se3_exp(dx, R, p);
        se3_act_inv(R, p, A6);
        se3_jexp(dx, B6);
        memset(nd->Fx, 0, sizeof(double) * ndx * ndx);
        memset(nd->Fu, 0, sizeof(double) * ndx * nv);
        for (int i = 0; i < ndx; ++i) nd->Fx[(long)i * ndx + i] = 1.0;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                nd->Fx[(long)i * ndx + j] = A6[6 * i + j];
                nd->Fx[(long)i * ndx + nv + j] = dt * B6[6 * i + j];
                nd->Fu[(long)i * nv + j] = dt * dt * B6[6 * i + j];
            }
        for (int i = 6; i < nv; ++i) { nd->Fx[(long)i * ndx + nv + i] = dt; nd->Fu[(long)i * nv + i] = dt * dt; }
        for (int i = 0; i < nv; ++i) nd->Fu[(long)(nv + i) * nv + i] = dt;
        for (int i = 0; i < ndx; ++i) nd->Lx[i] *= dt;
        for (long i = 0; i < (long)ndx * ndx; ++i) nd->Lxx[i] *= dt;
        for (int i = 0; i < nv; ++i) nd->Lu[i] *= dt;
        for (long i = 0; i < (long)nv * nv; ++i) nd->Luu[i] *= dt;
        nd->cost = dt * cost;
        memcpy(nd->xnext, xnext, sizeof(double) * (nq + nv));
    }
    return dt * cost;
}

/* ----------------------------------------------------------------------- SolverDDP --- */
/* C = A^T B,  A: n x p, B: n x q (row-major) */
static void gemm_tn(int n, int p, int q, const double *restrict A, const double *restrict B, double *restrict C) {
    memset(C, 0, sizeof(double) * p * q);
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < p; ++i) {
            const double a = A[(long)k * p + i];
            if (a == 0.0) continue;
            double *c = C + (long)i * q;
            const double *b = B + (long)k * q;
            for (int j = 0; j < q; ++j) c[j] += a * b[j];
        }
}
/* C = A B,  A: n x p, B: p x q */
static void gemm_nn(int n, int p, int q, const double *restrict A, const double *restrict B, double *restrict C) {
    memset(C, 0, sizeof(double) * n * q);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < p; ++k) {
            const double a = A[(long)i * p + k];
            if (a == 0.0) continue;
            double *c = C + (long)i * q;
            const double *b = B + (long)k * q;
            for (int j = 0; j < q; ++j) c[j] += a * b[j];
        }
}
static int all_finite(const double *a, long n) {
    for (long i = 0; i < n; ++i) if (!(fabs(a[i]) < INFINITY)) return 0;
    return 1;
}
/* lower Cholesky in place; 0 = not positive definite (Eigen::LLT info() != Success) */
static int cholesky(int n, double *A) {
    for (int j = 0; j < n; ++j) {
        double d = A[(long)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(long)j * n + k] * A[(long)j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[(long)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(long)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(long)i * n + k] * A[(long)j * n + k];
            A[(long)i * n + j] = s / d;
        }
    }
    return 1;
}
/* solve (L L^T) X = B in place, B: n x q */
static void chol_solve(int n, int q, const double *L, double *B) {
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < i; ++k) { const double l = L[(long)i * n + k]; for (int j = 0; j < q; ++j) B[(long)i * q + j] -= l * B[(long)k * q + j]; }
        const double d = L[(long)i * n + i];
        for (int j = 0; j < q; ++j) B[(long)i * q + j] /= d;
    }
    for (int i = n - 1; i >= 0; --i) {
        for (int k = i + 1; k < n; ++k) { const double l = L[(long)k * n + i]; for (int j = 0; j < q; ++j) B[(long)i * q + j] -= l * B[(long)k * q + j]; }
        const double d = L[(long)i * n + i];
        for (int j = 0; j < q; ++j) B[(long)i * q + j] /= d;
    }
}

typedef struct {
    int iters, status;          /* status: 0 converged, 1 maxiter reached, 2 regularisation hit reg_max */
    double cost, stop, reg;
} ikor_result_t;

/* per-iteration trace (optional, 4 doubles per iteration): cost after the iteration, regularisation after it,
 * accepted step length (0 = none accepted), stopping criterion */
#define IKOR_TRACE 4

/*
 * crocoddyl 1.9.0 SolverDDP::solve(init_xs = {}, init_us = {}, maxiter, is_feasible = false, reg_init = NaN):
 * cold start at the state's zero with gaps, regularisation x/10 within [1e-9, 1e9], step lengths 2^-k (k < 10),
 * accept when dVexp >= 0 and (d1 < th_grad || !feasible || dV > 0.1 dVexp), decrease regularisation when the step
 * length exceeds 0.5, increase it when <= 0.01, stop once feasible with |Q_u|^2 < 1e-9.
 */
static void solve_ddp(const ikor_problem_t *pb, const double *x0, int maxiter, double *xs_out, double *us_out,
                      ikor_result_t *res, double *trace) {
    const ikor_model_t *m = pb->m;
    const int nj = m->nj, nq = 7 + nj, nv = 6 + nj, ndx = 2 * nv, nx = nq + nv, T = pb->T;
    const double reg_min = 1e-9, reg_max = 1e9, th_grad = 1e-12, th_gaptol = 1e-16, th_stepdec = 0.5, th_stepinc = 0.01,
                 th_accept = 0.1, th_stop = 1e-9;
    node_t *nd = (node_t *)malloc(sizeof(node_t) * (size_t)(T + 1));
    double *xs = (double *)calloc((size_t)(T + 1) * nx, sizeof(double)), *us = (double *)calloc((size_t)T * nv, sizeof(double));
    double *xt = (double *)calloc((size_t)(T + 1) * nx, sizeof(double)), *ut = (double *)calloc((size_t)T * nv, sizeof(double));
    double *fs = (double *)calloc((size_t)(T + 1) * ndx, sizeof(double));
    double *K = (double *)calloc((size_t)T * nv * ndx, sizeof(double)), *kff = (double *)calloc((size_t)T * nv, sizeof(double));
    double *Qu = (double *)calloc((size_t)T * nv, sizeof(double)), *Quuk = (double *)calloc((size_t)T * nv, sizeof(double));
    double *Vxx = (double *)malloc(sizeof(double) * ndx * ndx), *Vx = (double *)malloc(sizeof(double) * ndx);
    double *FxTV = (double *)malloc(sizeof(double) * ndx * ndx), *Qxx = (double *)malloc(sizeof(double) * ndx * ndx);
    double *FuTV = (double *)malloc(sizeof(double) * nv * ndx), *Qxu = (double *)malloc(sizeof(double) * ndx * nv);
    double *Quu = (double *)malloc(sizeof(double) * nv * nv), *Lq = (double *)malloc(sizeof(double) * nv * nv);
    double *tmpxx = (double *)malloc(sizeof(double) * ndx * ndx);
    for (int t = 0; t <= T; ++t) xs[(long)t * nx + 6] = 1.0;          /* state zero: neutral q, v = 0; us = 0 */
    memcpy(xt, x0, sizeof(double) * nx);                              /* xs_try[0] = x0 */
    int feasible = 0, was_feasible = 0, recalc = 1, it_done = 0, status = 1;
    double xreg = reg_min, cost = 0.0, stop = INFINITY, d1 = 0.0, d2 = 0.0;

    for (int it = 0; it < maxiter; ++it) {
        it_done = it + 1;
        int gave_up = 0;
        for (;;) {      /* calcDiff (when the candidate changed) + backwardPass, retried with more regularisation on failure */
            if (recalc) {
                cost = 0.0;
                for (int t = 0; t <= T; ++t) {
                    double xn[MAXJ + 7 + MAXV];
                    cost += node_calc(pb, t, xs + (long)t * nx, t < T ? us + (long)t * nv : NULL, xn, &nd[t]);
                }
                if (!feasible) {
                    int ok = 1;
                    for (int t = 0; t <= T; ++t) {
                        state_diff(nj, xs + (long)t * nx, t == 0 ? x0 : nd[t - 1].xnext, fs + (long)t * ndx, NULL);
                        for (int i = 0; i < ndx; ++i) if (!(fabs(fs[(long)t * ndx + i]) < th_gaptol)) ok = 0;
                    }
                    feasible = ok;
                } else if (!was_feasible) memset(fs, 0, sizeof(double) * (size_t)(T + 1) * ndx);
            }
            /* backwardPass */
            int bad = 0;
            memcpy(Vxx, nd[T].Lxx, sizeof(double) * ndx * ndx);
            memcpy(Vx, nd[T].Lx, sizeof(double) * ndx);
            for (int i = 0; i < ndx; ++i) Vxx[(long)i * ndx + i] += xreg;
            if (!feasible)
                for (int i = 0; i < ndx; ++i) { double s = 0.0; for (int j = 0; j < ndx; ++j) s += Vxx[(long)i * ndx + j] * fs[(long)T * ndx + j]; Vx[i] += s; }
            for (int t = T - 1; t >= 0 && !bad; --t) {
                const node_t *d = &nd[t];
                gemm_tn(ndx, ndx, ndx, d->Fx, Vxx, FxTV);
                gemm_nn(ndx, ndx, ndx, FxTV, d->Fx, Qxx);
                for (long i = 0; i < (long)ndx * ndx; ++i) Qxx[i] += d->Lxx[i];
                gemm_nn(ndx, ndx, nv, FxTV, d->Fu, Qxu);
                gemm_tn(ndx, nv, ndx, d->Fu, Vxx, FuTV);
                gemm_nn(nv, ndx, nv, FuTV, d->Fu, Quu);
                for (long i = 0; i < (long)nv * nv; ++i) Quu[i] += d->Luu[i];
                double Qx[MAXDX], *qu = Qu + (long)t * nv;
                for (int i = 0; i < ndx; ++i) { double s = 0.0; for (int k = 0; k < ndx; ++k) s += d->Fx[(long)k * ndx + i] * Vx[k]; Qx[i] = d->Lx[i] + s; }
                for (int i = 0; i < nv; ++i) { double s = 0.0; for (int k = 0; k < ndx; ++k) s += d->Fu[(long)k * nv + i] * Vx[k]; qu[i] = d->Lu[i] + s; }
                for (int i = 0; i < nv; ++i) Quu[(long)i * nv + i] += xreg;
                memcpy(Lq, Quu, sizeof(double) * nv * nv);
                if (!cholesky(nv, Lq)) { bad = 1; break; }
                /* K = Quu^-1 Qux, k = Quu^-1 Qu */
                double *Kt = K + (long)t * nv * ndx, *kt = kff + (long)t * nv;
                for (int i = 0; i < nv; ++i) for (int j = 0; j < ndx; ++j) Kt[(long)i * ndx + j] = Qxu[(long)j * nv + i];
                chol_solve(nv, ndx, Lq, Kt);
                memcpy(kt, qu, sizeof(double) * nv);
                chol_solve(nv, 1, Lq, kt);
                for (int i = 0; i < nv; ++i) { double s = 0.0; for (int j = 0; j < nv; ++j) s += Quu[(long)i * nv + j] * kt[j]; Quuk[(long)t * nv + i] = s; }
                /* Vx = Qx - K^T Qu;  Vxx = Qxx - Qxu K, symmetrised, + xreg I;  Vx += Vxx fs[t] while infeasible */
                for (int i = 0; i < ndx; ++i) { double s = 0.0; for (int p = 0; p < nv; ++p) s += Kt[(long)p * ndx + i] * qu[p]; Vx[i] = Qx[i] - s; }
                gemm_nn(ndx, nv, ndx, Qxu, Kt, tmpxx);
                for (long i = 0; i < (long)ndx * ndx; ++i) Qxx[i] -= tmpxx[i];
                for (int i = 0; i < ndx; ++i)
                    for (int j = 0; j < ndx; ++j) Vxx[(long)i * ndx + j] = 0.5 * (Qxx[(long)i * ndx + j] + Qxx[(long)j * ndx + i]);
                for (int i = 0; i < ndx; ++i) Vxx[(long)i * ndx + i] += xreg;
                if (!feasible)
                    for (int i = 0; i < ndx; ++i) { double s = 0.0; for (int j = 0; j < ndx; ++j) s += Vxx[(long)i * ndx + j] * fs[(long)t * ndx + j]; Vx[i] += s; }
                if (!all_finite(Vx, ndx) || !all_finite(Vxx, (long)ndx * ndx)) bad = 1;
            }
            if (!bad) break;
            recalc = 0;
            xreg = fmin(xreg * 10.0, reg_max);
            if (xreg == reg_max) { gave_up = 1; break; }
        }
        if (gave_up) { status = 2; break; }
        /* expectedImprovement */
        d1 = 0.0; d2 = 0.0;
        for (long i = 0; i < (long)T * nv; ++i) { d1 += Qu[i] * kff[i]; d2 -= kff[i] * Quuk[i]; }
        recalc = 0;
        double alpha = 0.0, alpha_acc = 0.0;
        for (int ia = 0; ia < 10; ++ia) {
            alpha = ldexp(1.0, -ia);
            /* forwardPass(alpha) */
            double ctry = 0.0;
            int threw = 0;
            for (int t = 0; t < T && !threw; ++t) {
                double dx[MAXDX];
                state_diff(nj, xs + (long)t * nx, xt + (long)t * nx, dx, NULL);
                const double *Kt = K + (long)t * nv * ndx;
                double *u = ut + (long)t * nv;
                for (int i = 0; i < nv; ++i) {
                    double s = us[(long)t * nv + i] - alpha * kff[(long)t * nv + i];
                    for (int j = 0; j < ndx; ++j) s -= Kt[(long)i * ndx + j] * dx[j];
                    u[i] = s;
                }
                ctry += node_calc(pb, t, xt + (long)t * nx, u, xt + (long)(t + 1) * nx, NULL);
                if (!(fabs(ctry) < INFINITY) || !all_finite(xt + (long)(t + 1) * nx, nx)) threw = 1;
            }
            if (threw) continue;
            ctry += node_calc(pb, T, xt + (long)T * nx, NULL, NULL, NULL);
            if (!(fabs(ctry) < INFINITY)) continue;
            const double dV = cost - ctry, dVexp = alpha * (d1 + 0.5 * alpha * d2);
            if (dVexp >= 0.0 && (d1 < th_grad || !feasible || dV > th_accept * dVexp)) {
                was_feasible = feasible;
                memcpy(xs, xt, sizeof(double) * (size_t)(T + 1) * nx);
                memcpy(us, ut, sizeof(double) * (size_t)T * nv);
                feasible = 1;
                cost = ctry;
                recalc = 1;
                alpha_acc = alpha;
                break;
            }
        }
        if (alpha > th_stepdec) xreg = fmax(xreg / 10.0, reg_min);
        int hit_max = 0;
        if (alpha <= th_stepinc) {
            xreg = fmin(xreg * 10.0, reg_max);
            if (xreg == reg_max) hit_max = 1;
        }
        if (!hit_max) { stop = 0.0; for (long i = 0; i < (long)T * nv; ++i) stop += Qu[i] * Qu[i]; }
        if (trace) { double *tr = trace + (long)it * IKOR_TRACE; tr[0] = cost; tr[1] = xreg; tr[2] = alpha_acc; tr[3] = stop; }
        if (hit_max) { status = 2; break; }
        if (was_feasible && stop < th_stop) { status = 0; break; }
    }
    memcpy(xs_out, xs, sizeof(double) * (size_t)(T + 1) * nx);
    memcpy(us_out, us, sizeof(double) * (size_t)T * nv);
    res->iters = it_done; res->status = status; res->cost = cost; res->stop = stop; res->reg = xreg;
    free(nd); free(xs); free(us); free(xt); free(ut); free(fs); free(K); free(kff); free(Qu); free(Quuk);
    free(Vxx); free(Vx); free(FxTV); free(Qxx); free(FuTV); free(Qxu); free(Quu); free(Lq); free(tmpxx);
}

/* ------------------------------------------------------------------------ exported --- */
void ikor_model_fill(ikor_model_t *m, int nj, const int *parent, const double *R, const double *p, const double *axis,
                     const double *mass, const double *com, const double *inertia, int nframes, const int *frame_body,
                     const double *frame_p) {
    memset(m, 0, sizeof *m);
    m->nj = nj; m->nframes = nframes;
    for (int i = 0; i < nj; ++i) {
        m->parent[i] = parent[i];
        memcpy(m->R[i], R + 9 * i, sizeof(double) * 9);
        memcpy(m->p[i], p + 3 * i, sizeof(double) * 3);
        memcpy(m->axis[i], axis + 3 * i, sizeof(double) * 3);
    }
    for (int b = 0; b <= nj; ++b) {
        m->mass[b] = mass[b];
        memcpy(m->com[b], com + 3 * b, sizeof(double) * 3);
        memcpy(m->inertia[b], inertia + 9 * b, sizeof(double) * 9);
    }
    for (int f = 0; f < nframes; ++f) { m->frame_body[f] = frame_body[f]; memcpy(m->frame_p[f], frame_p + 3 * f, sizeof(double) * 3); }
}
int ikor_model_bytes(void) { return (int)sizeof(ikor_model_t); }

/* B independent InverseKinematics::optimize calls, the arrays the GPU batch takes (bmpc_ik_batch_t): x0 [B][nx],
 * dt [B][T], tasks [B][T+1][33], state_w / x_reg / ctrl_w with batch strides s_* and node strides sn_*.
 * Outputs: xs [B][T+1][nx], us [B][T][nv], iters/status [B], cost/stop/reg [B]; trace (or NULL) [B][maxiter][4].
 * One problem per OpenMP thread (nthreads <= 0: all). */
void ikor_solve_batch(const ikor_model_t *m, int B, int T, int maxiter, const double *x0, const double *dt, const double *tasks,
                      const double *state_w, long s_sw, long sn_sw, const double *x_reg, long s_xr, long sn_xr,
                      const double *ctrl_w, long s_cw, long sn_cw, double *xs, double *us, int *iters, int *status,
                      double *cost, double *stop, double *reg, double *trace, int nthreads) {
    const int nq = 7 + m->nj, nv = 6 + m->nj, nx = nq + nv;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int b = 0; b < B; ++b) {
        ikor_problem_t pb;
        pb.m = m; pb.T = T; pb.dt = dt + (long)b * T; pb.tasks = tasks + (long)b * (T + 1) * NTASK;
        pb.state_w = state_w + s_sw * b; pb.x_reg = x_reg + s_xr * b; pb.ctrl_w = ctrl_w + s_cw * b;
        pb.sn_sw = sn_sw; pb.sn_xr = sn_xr; pb.sn_cw = sn_cw;
        ikor_result_t r;
        solve_ddp(&pb, x0 + (long)b * nx, maxiter, xs + (long)b * (T + 1) * nx, us + (long)b * T * nv, &r,
                  trace ? trace + (long)b * maxiter * IKOR_TRACE : NULL);
        iters[b] = r.iters; status[b] = r.status; cost[b] = r.cost; stop[b] = r.stop; reg[b] = r.reg;
    }
    (void)nthreads;
}

/* centroidal state [com, vcom, L] of (q, v): KinoDynMP::optimize's x0 for the centroidal solve (kino_dyn.cpp:42,86-97) */
void ikor_centroidal_state(const ikor_model_t *m, int B, const double *x, double *out9) {
    const int nq = 7 + m->nj, nx = nq + 6 + m->nj;
    for (int b = 0; b < B; ++b) {
        kin_t k;
        kin_compute(m, x + (long)b * nx, x + (long)b * nx + nq, &k);
        for (int a = 0; a < 3; ++a) { out9[9 * b + a] = k.com[a]; out9[9 * b + 3 + a] = k.hg[a] / k.M; out9[9 * b + 6 + a] = k.hg[3 + a]; }
    }
}

/* pieces exported for the CPU tests (finite-difference pins and comparison with oracle/rbd_np.py) */
void ikor_kin_quantities(const ikor_model_t *m, const double *x, double *com, double *hg, double *Ag, double *Jc, double *dh) {
    const int nq = 7 + m->nj;
    kin_t k;
    kin_compute(m, x, x + nq, &k);
    memcpy(com, k.com, sizeof k.com);
    memcpy(hg, k.hg, sizeof k.hg);
    centroidal_derivs(m, &k, Ag, Jc, dh);
}
void ikor_frame(const ikor_model_t *m, const double *x, int f, double *pos, double *J) {
    const int nq = 7 + m->nj;
    kin_t k;
    kin_compute(m, x, x + nq, &k);
    frame_position(m, &k, f, pos);
    frame_jacobian(m, &k, f, J);
}
void ikor_state_ops(const ikor_model_t *m, const double *x0, const double *x1, const double *dx, double *diff, double *Jl, double *xint,
                    double *A6, double *B6) {
    state_diff(m->nj, x0, x1, diff, Jl);
    state_integrate(m->nj, x0, dx, xint);
    double R[9], p[3];
    se3_exp(dx, R, p);
    se3_act_inv(R, p, A6);
    se3_jexp(dx, B6);
}
/* cost and dense derivatives of one node (for the derivative checks) */
double ikor_node(const ikor_model_t *m, int T, int t, const double *dt, const double *tasks, const double *state_w, const double *x_reg,
                 const double *ctrl_w, const double *x, const double *u, double *xnext, double *Lx, double *Lxx, double *Lu, double *Luu,
                 double *Fx, double *Fu) {
    ikor_problem_t pb = {m, T, dt, tasks, state_w, x_reg, ctrl_w, 0, 0, 0};
    const int nv = 6 + m->nj, ndx = 2 * nv;
    node_t *nd = (node_t *)malloc(sizeof(node_t));
    double xn[MAXJ + 7 + MAXV];
    const double c = node_calc(&pb, t, x, u, xn, nd);
    if (t < T && xnext) memcpy(xnext, xn, sizeof(double) * (7 + m->nj + nv));
    memcpy(Lx, nd->Lx, sizeof(double) * ndx);
    memcpy(Lxx, nd->Lxx, sizeof(double) * ndx * ndx);
    if (t < T) {
        memcpy(Lu, nd->Lu, sizeof(double) * nv);
        memcpy(Luu, nd->Luu, sizeof(double) * nv * nv);
        memcpy(Fx, nd->Fx, sizeof(double) * ndx * ndx);
        memcpy(Fu, nd->Fu, sizeof(double) * ndx * nv);
    }
    free(nd);
    return c;
}

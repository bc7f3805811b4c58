/*
 * oracle/biconvex_fast.c -- a second CPU baseline for bench.py: the same ADMM / FISTA iteration as
 * biconvex_oracle.c, but MATRIX-FREE (per-knot operators, no sparse matrices, no Hessian rebuild, no
 * allocation inside the solve) -- i.e. what a CPU programmer would write once the structure of A_x / A_f
 * is exploited, the same exact-algebra restructurings the GPU kernel uses (acceptance test as
 * d'Qd + rho|Ad|^2 > (L/2)|d|^2, x_init rows folded into knot 0's diagonal cost, A-images carried through
 * the momentum step by linearity).  SURVEY.md 8d asks for the GPU speed-up to be quoted against this
 * faster CPU variant as well as against the reference's own formulation.
 *
 * TEST INFRASTRUCTURE ONLY (see biconvex_oracle.h): used by tests/ (it must agree with the strict
 * restatement) and by bench.py's cpu_baseline leg.  PARITY UNPINNED like the rest of oracle/.
 *
 * Reference behaviour restated: the files listed in biconvex_oracle.h; raw cost form (Qx, qx, lbx, ubx, Qf, qf).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "biconvex_oracle.h"

#define GRAV 9.81
#define MAXE 4

typedef struct {
    int H, E, maxit;
    double m, rho, beta, mu, tol, exit_tol;
    const double *cnt, *dt, *x_init, *Qx, *qx, *Qf, *qf, *lbx, *ubx;
    double *cm;                         /* momentum coefficients (t_k - 1)/t_{k+1}, fista.cpp:34-35 */
    /* work arrays, sized for the larger of the two problems */
    double *xa, *xb, *y, *ra, *rb, *ry, *g, *bpk, *an, *sp, *S, *bf, *qd, *q;
} fast_t;

/* ---------------------------------------------------------------- force step (centroidal.cpp:57-84) */
static void applyA_f(const fast_t *w, const double *v, double *u) {   /* u = A_x v + bPk on rows 9t+3..8 -> 6 per knot */
    const int H = w->H, E = w->E;
    for (int t = 0; t < H; ++t) {
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (int n = 0; n < E; ++n) {
            const double a = w->an[t * E + n], *sp = w->sp + (t * E + n) * 3, *f = v + (t * E + n) * 3;
            s[0] += a * f[0]; s[1] += a * f[1]; s[2] += a * f[2];
            s[3] += sp[2] * f[1] - sp[1] * f[2];
            s[4] += sp[0] * f[2] - sp[2] * f[0];
            s[5] += sp[1] * f[0] - sp[0] * f[1];
        }
        for (int k = 0; k < 6; ++k) u[6 * t + k] = s[k] + w->bpk[6 * t + k];
    }
}

static void project_f(const fast_t *w, double *f) {                    /* fista.cpp:52-70, as written there */
    const int n = w->H * w->E;
    const double mu = w->mu;
    for (int i = 0; i < n; ++i) {
        double *p = f + 3 * i;
        const double s = p[0] * p[0] + p[1] * p[1], z = p[2];
        if (s * mu < -z || z < 0) { p[0] = p[1] = p[2] = 0.0; }
        else if (s > mu * z) {
            const double k = (mu * mu * s + mu * z) / ((mu * mu + 1.0) * s);
            p[0] *= k; p[1] *= k; p[2] = (mu * s + z) / (mu * mu + 1.0);
        }
    }
}

/* one FISTA::optimize (fista.cpp:29-50) on the force problem; x in/out; returns iterations, adds retries */
static int fista_f(fast_t *w, double *x, double *L, int *retries) {
    const int H = w->H, E = w->E, nf = 3 * E * H, nr = 6 * H;
    double *xo = w->xa, *xn = w->xb, *y = w->y, *ro = w->ra, *rn = w->rb, *ry = w->ry, *g = w->g;
    memcpy(xo, x, sizeof(double) * nf); memcpy(y, x, sizeof(double) * nf);
    applyA_f(w, y, ry); memcpy(ro, ry, sizeof(double) * nr);
    int it = 0;
    for (int i = 0; i < w->maxit; ++i) {
        /* g = 2 Q y + q + 2 rho A^T (A y + bPk)   (problem.cpp:36-38,54-56) */
        for (int t = 0; t < H; ++t)
            for (int n = 0; n < E; ++n) {
                const double a = w->an[t * E + n], *sp = w->sp + (t * E + n) * 3, *r = ry + 6 * t;
                const int j = (t * E + n) * 3;
                const double zx = a * r[0] - sp[2] * r[4] + sp[1] * r[5];
                const double zy = a * r[1] + sp[2] * r[3] - sp[0] * r[5];
                const double zz = a * r[2] - sp[1] * r[3] + sp[0] * r[4];
                g[j] = 2.0 * w->Qf[j] * y[j] + 2.0 * w->rho * zx;
                g[j + 1] = 2.0 * w->Qf[j + 1] * y[j + 1] + 2.0 * w->rho * zy;
                g[j + 2] = 2.0 * w->Qf[j + 2] * y[j + 2] + 2.0 * w->rho * zz;
                if (w->qf) { g[j] += w->qf[j]; g[j + 1] += w->qf[j + 1]; g[j + 2] += w->qf[j + 2]; }
            }
        double Gn;
        for (;;) {                                                     /* fista.cpp:8-26 */
            for (int j = 0; j < nf; ++j) xn[j] = y[j] - g[j] / *L;
            project_f(w, xn);
            applyA_f(w, xn, rn);
            double g2 = 0, cv = 0, e2 = 0;
            for (int j = 0; j < nf; ++j) { const double d = xn[j] - y[j]; g2 += d * d; cv += w->Qf[j] * d * d; }
            for (int k = 0; k < nr; ++k) { const double e = rn[k] - ry[k]; e2 += e * e; }
            cv += w->rho * e2;
            Gn = sqrt(g2);
            if (cv > (*L * 0.5) * (Gn * Gn)) { *L *= w->beta; ++*retries; } else break;
        }
        ++it;
        const double c = w->cm[i];
        for (int j = 0; j < nf; ++j) y[j] = xn[j] + c * (xn[j] - xo[j]);
        for (int k = 0; k < nr; ++k) ry[k] = rn[k] + c * (rn[k] - ro[k]);
        double *tp = xo; xo = xn; xn = tp; tp = ro; ro = rn; rn = tp;   /* x_k = x_k_1 */
        if (Gn < w->tol) break;
    }
    memcpy(x, xo, sizeof(double) * nf);
    return it;
}

/* --------------------------------------------------------------- motion step (centroidal.cpp:86-127) */
static void applyA_x(const fast_t *w, const double *v, double *u) {   /* u = A_f v + bPk on row-blocks t < H */
    const int H = w->H;
    for (int t = 0; t < H; ++t) {
        const double *a = v + 9 * t, *b = v + 9 * (t + 1), *S = w->S + 3 * t, dt = w->dt[t];
        double r[9];
        for (int l = 0; l < 9; ++l) r[l] = a[l] - b[l];
        for (int k = 0; k < 3; ++k) r[k] += dt * b[3 + k];
        r[6] += S[1] * a[2] - S[2] * a[1];
        r[7] += S[2] * a[0] - S[0] * a[2];
        r[8] += S[0] * a[1] - S[1] * a[0];
        for (int l = 0; l < 9; ++l) u[9 * t + l] = r[l] + w->bpk[9 * t + l];
    }
}

static int fista_x(fast_t *w, double *x, double *L, int *retries) {
    const int H = w->H, nx = 9 * (H + 1), nr = 9 * H;
    double *xo = w->xa, *xn = w->xb, *y = w->y, *ro = w->ra, *rn = w->rb, *ry = w->ry, *g = w->g;
    memcpy(xo, x, sizeof(double) * nx); memcpy(y, x, sizeof(double) * nx);
    applyA_x(w, y, ry); memcpy(ro, ry, sizeof(double) * nr);
    int it = 0;
    for (int i = 0; i < w->maxit; ++i) {
        for (int t = 0; t <= H; ++t) {                                 /* A_f^T (A_f y + bPk), block column t */
            double z[9];
            const double *rt = t < H ? ry + 9 * t : NULL, *rp = t > 0 ? ry + 9 * (t - 1) : NULL;
            for (int l = 0; l < 9; ++l) z[l] = (rt ? rt[l] : 0.0) - (rp ? rp[l] : 0.0);
            if (rp) for (int k = 0; k < 3; ++k) z[3 + k] += w->dt[t - 1] * rp[k];
            if (rt) {
                const double *S = w->S + 3 * t;
                z[0] += S[2] * rt[7] - S[1] * rt[8];
                z[1] += S[0] * rt[8] - S[2] * rt[6];
                z[2] += S[1] * rt[6] - S[0] * rt[7];
            }
            for (int l = 0; l < 9; ++l) g[9 * t + l] = 2.0 * w->qd[9 * t + l] * y[9 * t + l] + (2.0 * w->rho * z[l] + w->q[9 * t + l]);
        }
        double Gn;
        for (;;) {
            for (int j = 0; j < nx; ++j) {
                double v = y[j] - g[j] / *L;
                v = v > w->ubx[j] ? w->ubx[j] : v;
                xn[j] = v < w->lbx[j] ? w->lbx[j] : v;                  /* fista.cpp:10 */
            }
            applyA_x(w, xn, rn);
            double g2 = 0, cv = 0, e2 = 0;
            for (int j = 0; j < nx; ++j) { const double d = xn[j] - y[j]; g2 += d * d; cv += w->qd[j] * d * d; }
            for (int k = 0; k < nr; ++k) { const double e = rn[k] - ry[k]; e2 += e * e; }
            cv += w->rho * e2;
            Gn = sqrt(g2);
            if (cv > (*L * 0.5) * (Gn * Gn)) { *L *= w->beta; ++*retries; } else break;
        }
        ++it;
        const double c = w->cm[i];
        for (int j = 0; j < nx; ++j) y[j] = xn[j] + c * (xn[j] - xo[j]);
        for (int k = 0; k < nr; ++k) ry[k] = rn[k] + c * (rn[k] - ro[k]);
        double *tp = xo; xo = xn; xn = tp; tp = ro; ro = rn; rn = tp;
        if (Gn < w->tol) break;
    }
    memcpy(x, xo, sizeof(double) * nx);
    return it;
}

/* ----------------------------------------------------------------------------- ADMM (biconvex.cpp:80-120) */
static int solve_one(fast_t *w, double *X, double *F, double *P, double *L_x, double *L_f, int num_iters, int *stats, double *hist,
                     int *trace) {
    const int H = w->H, E = w->E, nx = 9 * (H + 1);
    int n_admm = 0, it_f = 0, it_x = 0, bt_f = 0, bt_x = 0, status = 0;
    for (int it = 0; it < num_iters; ++it) {
        /* ---- force step: A_x(X), bPk = -b_x + P */
        for (int t = 0; t < H; ++t) {
            const double dt = w->dt[t];
            for (int n = 0; n < E; ++n) {
                const double *c = w->cnt + (t * E + n) * 4;
                w->an[t * E + n] = c[0] * (dt / w->m);
                for (int k = 0; k < 3; ++k) w->sp[(t * E + n) * 3 + k] = c[0] * (X[9 * t + k] - c[1 + k]) * dt;
            }
            for (int k = 0; k < 6; ++k) {
                double bx = X[9 * (t + 1) + 3 + k] - X[9 * t + 3 + k];
                if (k == 2) bx += GRAV * dt;
                w->bpk[6 * t + k] = -bx + P[9 * t + 3 + k];
            }
        }
        it_f += fista_f(w, F, L_f, &bt_f);
        /* ---- motion step: A_f(F), b_f; x_init rows folded into knot 0 */
        for (int t = 0; t < H; ++t) {
            const double dt = w->dt[t];
            double S[3] = {0, 0, 0}, b[6] = {0, 0, 0, 0, 0, 0};
            for (int n = 0; n < E; ++n) {
                const double *c = w->cnt + (t * E + n) * 4, *f = F + (t * E + n) * 3;
                S[0] += c[0] * f[0] * dt; S[1] += c[0] * f[1] * dt; S[2] += c[0] * f[2] * dt;
                b[0] += -c[0] * f[0] * dt / w->m; b[1] += -c[0] * f[1] * dt / w->m; b[2] += -c[0] * f[2] * dt / w->m;
                b[3] += (c[0] * f[1] * c[3] - c[0] * f[2] * c[2]) * dt;
                b[4] += (c[0] * f[2] * c[1] - c[0] * f[0] * c[3]) * dt;
                b[5] += (c[0] * f[0] * c[2] - c[0] * f[1] * c[1]) * dt;
            }
            for (int k = 0; k < 3; ++k) w->S[3 * t + k] = S[k];
            double *bf = w->bf + 9 * t;
            bf[0] = bf[1] = bf[2] = 0.0;
            bf[3] = b[0]; bf[4] = b[1]; bf[5] = b[2] + GRAV * dt; bf[6] = b[3]; bf[7] = b[4]; bf[8] = b[5];
            for (int l = 0; l < 9; ++l) w->bpk[9 * t + l] = -bf[l] + P[9 * t + l];
        }
        for (int j = 0; j < nx; ++j) { w->qd[j] = w->Qx[j]; w->q[j] = w->qx[j]; }
        for (int l = 0; l < 9; ++l) {   /* rho |X_0 + (P_H - x_init)|^2   (centroidal.hpp:22-27) */
            w->qd[l] += w->rho;
            w->q[l] += 2.0 * w->rho * (P[9 * H + l] - w->x_init[l]);
        }
        it_x += fista_x(w, X, L_x, &bt_x);
        /* ---- dyn_violation = A_f X - b_f ; P += dyn_violation */
        double v2 = 0.0;
        for (int t = 0; t < H; ++t) {
            const double *a = X + 9 * t, *b = X + 9 * (t + 1), *S = w->S + 3 * t, dt = w->dt[t];
            double r[9];
            for (int l = 0; l < 9; ++l) r[l] = a[l] - b[l];
            for (int k = 0; k < 3; ++k) r[k] += dt * b[3 + k];
            r[6] += S[1] * a[2] - S[2] * a[1];
            r[7] += S[2] * a[0] - S[0] * a[2];
            r[8] += S[0] * a[1] - S[1] * a[0];
            for (int l = 0; l < 9; ++l) { const double d = r[l] - w->bf[9 * t + l]; P[9 * t + l] += d; v2 += d * d; }
        }
        for (int l = 0; l < 9; ++l) { const double d = X[l] - w->x_init[l]; P[9 * H + l] += d; v2 += d * d; }
        const double nrm = sqrt(v2);
        ++n_admm;
        if (hist) hist[it] = nrm;
        if (trace) { trace[4 * it] = it_f; trace[4 * it + 1] = it_x; trace[4 * it + 2] = bt_f; trace[4 * it + 3] = bt_x; }
        if (isnan(nrm)) { status = 2; break; }
        if (nrm < w->exit_tol) break;
    }
    stats[0] = n_admm; stats[1] = it_f; stats[2] = it_x; stats[3] = bt_f; stats[4] = bt_x; stats[5] = status;
    return status;
}

int orc_fast_solve_batch_traced(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                                const double *cnt_plan, const double *dt, const double *x_init,
                                const double *Qx, const double *qx, const double *Qf, const double *qf,
                                const double *lbx, const double *ubx, int shared_cost,
                                double *X, double *F, double *P, double *L_x, double *L_f, int num_iters,
                                int *stats, int nthreads, double *hist, int *trace) {
    const int H = n_col, E = n_eff, nx = 9 * (H + 1), nf = 3 * E * H;
    if (E > MAXE) return -1;
    const int nmax = nx > nf ? nx : nf;
    int ndiv = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel reduction(+ : ndiv)
    {
        fast_t w;
        w.H = H; w.E = E; w.maxit = prm->maxit; w.m = m; w.rho = prm->rho; w.beta = prm->beta; w.mu = prm->mu;
        w.tol = prm->tol; w.exit_tol = prm->exit_tol;
        double *buf = (double *)malloc(sizeof(double) * (size_t)(prm->maxit + 7 * nmax + 9 * H + E * H * 4 + 3 * H + 9 * H + 2 * nx + 64));
        double *p = buf;
        w.cm = p; p += prm->maxit;
        w.xa = p; p += nmax; w.xb = p; p += nmax; w.y = p; p += nmax; w.ra = p; p += nmax; w.rb = p; p += nmax;
        w.ry = p; p += nmax; w.g = p; p += nmax;
        w.bpk = p; p += 9 * H; w.an = p; p += E * H; w.sp = p; p += 3 * E * H; w.S = p; p += 3 * H; w.bf = p; p += 9 * H;
        w.qd = p; p += nx; w.q = p; p += nx;
        {   /* t+ = 1 + sqrt(1 + 4 t^2)/2 (sic, fista.cpp:34) */
            double tk = 1.0;
            for (int i = 0; i < prm->maxit; ++i) { const double t1 = 1.0 + sqrt(1.0 + 4.0 * tk * tk) * 0.5; w.cm[i] = (tk - 1.0) / t1; tk = t1; }
        }
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < B; ++b) {
            const long cb = shared_cost ? 0 : b;
            w.cnt = cnt_plan + (long)b * H * E * 4; w.dt = dt + (long)b * H; w.x_init = x_init + (long)b * 9;
            w.Qx = Qx + cb * nx; w.qx = qx + (long)b * nx; w.Qf = Qf + cb * nf; w.qf = qf ? qf + cb * nf : NULL;
            w.lbx = lbx + (long)b * nx; w.ubx = ubx + (long)b * nx;
            if (solve_one(&w, X + (long)b * nx, F + (long)b * nf, P + (long)b * nx, L_x + b, L_f + b, num_iters, stats + (long)b * ORC_NSTATS,
                          hist ? hist + (long)b * num_iters : NULL, trace ? trace + (long)b * num_iters * 4 : NULL))
                ++ndiv;
        }
        free(buf);
    }
    return ndiv;
}

int orc_fast_solve_batch(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                         const double *cnt_plan, const double *dt, const double *x_init,
                         const double *Qx, const double *qx, const double *Qf, const double *qf,
                         const double *lbx, const double *ubx, int shared_cost,
                         double *X, double *F, double *P, double *L_x, double *L_f, int num_iters,
                         int *stats, int nthreads) {
    return orc_fast_solve_batch_traced(B, n_col, n_eff, m, prm, cnt_plan, dt, x_init, Qx, qx, Qf, qf, lbx, ubx, shared_cost, X, F, P,
                                       L_x, L_f, num_iters, stats, nthreads, NULL, NULL);
}

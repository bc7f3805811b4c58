/*
 * oracle/biconvex_oracle.c -- see biconvex_oracle.h.  TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED (no reference golden vectors exist; reference unbuildable here).
 *
 * "Strict" restatement: same formulation as the reference -- explicit sparse
 * A_x / A_f, explicit Hessian ATA_ = 2(Q + rho A^T A) rebuilt by a sparse-sparse
 * product every ADMM iteration, gradient = ATA_ y + ATbPk_, objective difference
 * through two sparse mat-vecs.  Column-major (CSC) storage and accumulation
 * order chosen as Eigen's default SparseMatrix<double> would do it; reductions
 * (dot / norm) are plain left-to-right sums.  Compile with -ffp-contract=off.
 */
#include "biconvex_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ CSC --- */
typedef struct {
    int rows, cols, nnz;
    int *colptr; /* cols+1 */
    int *rowidx; /* nnz, sorted inside each column */
    double *val;
} csc_t;

static void csc_free(csc_t *a) {
    free(a->colptr); free(a->rowidx); free(a->val);
    memset(a, 0, sizeof(*a));
}

static void csc_alloc(csc_t *a, int rows, int cols, int nnz) {
    a->rows = rows; a->cols = cols; a->nnz = nnz;
    a->colptr = (int *)calloc((size_t)cols + 1, sizeof(int));
    a->rowidx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    a->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
}

/* triplets (unique (r,c)) -> CSC with sorted rows */
static void csc_from_triplets(csc_t *a, int rows, int cols, int nt,
                              const int *tr, const int *tc, const double *tv) {
    csc_alloc(a, rows, cols, nt);
    for (int k = 0; k < nt; ++k) a->colptr[tc[k] + 1]++;
    for (int j = 0; j < cols; ++j) a->colptr[j + 1] += a->colptr[j];
    int *fill = (int *)calloc((size_t)cols, sizeof(int));
    for (int k = 0; k < nt; ++k) {
        int j = tc[k];
        int p = a->colptr[j] + fill[j]++;
        /* insertion keeps rows sorted */
        while (p > a->colptr[j] && a->rowidx[p - 1] > tr[k]) {
            a->rowidx[p] = a->rowidx[p - 1];
            a->val[p] = a->val[p - 1];
            --p;
        }
        a->rowidx[p] = tr[k];
        a->val[p] = tv[k];
    }
    free(fill);
}

static void csc_transpose(const csc_t *a, csc_t *t) {
    csc_alloc(t, a->cols, a->rows, a->nnz);
    for (int p = 0; p < a->nnz; ++p) t->colptr[a->rowidx[p] + 1]++;
    for (int j = 0; j < t->cols; ++j) t->colptr[j + 1] += t->colptr[j];
    int *fill = (int *)calloc((size_t)t->cols, sizeof(int));
    for (int j = 0; j < a->cols; ++j)
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p) {
            int i = a->rowidx[p];
            int d = t->colptr[i] + fill[i]++;
            t->rowidx[d] = j; /* j increasing -> sorted */
            t->val[d] = a->val[p];
        }
    free(fill);
}

/* y = A x  (column-major axpy order: y_i accumulates over increasing j) */
static void csc_matvec(const csc_t *a, const double *x, double *y) {
    for (int i = 0; i < a->rows; ++i) y[i] = 0.0;
    for (int j = 0; j < a->cols; ++j) {
        const double xj = x[j];
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p)
            y[a->rowidx[p]] += a->val[p] * xj;
    }
}

/* y_j = sum_k A_kj x_k  (A^T x; one sequential dot per column) */
static void csc_tmatvec(const csc_t *a, const double *x, double *y) {
    for (int j = 0; j < a->cols; ++j) {
        double s = 0.0;
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p)
            s += a->val[p] * x[a->rowidx[p]];
        y[j] = s;
    }
}

/* H = 2 (diag(Qd) + rho A^T A): problem.cpp:36.  at = A^T in CSC. */
static void build_hessian(const csc_t *a, const csc_t *at, const double *Qd, double rho,
                          csc_t *h) {
    const int n = a->cols;
    double *acc = (double *)calloc((size_t)n, sizeof(double));
    int *mark = (int *)malloc(sizeof(int) * (size_t)n);
    int *list = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) mark[i] = -1;
    /* pass 1: count */
    int cap = 0;
    int *cnt = (int *)calloc((size_t)n, sizeof(int));
    for (int j = 0; j < n; ++j) {
        int c = 0;
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p) {
            int k = a->rowidx[p];
            for (int s = at->colptr[k]; s < at->colptr[k + 1]; ++s) {
                int i = at->rowidx[s];
                if (mark[i] != j) { mark[i] = j; ++c; }
            }
        }
        if (mark[j] != j) { mark[j] = j; ++c; } /* diagonal of Q */
        cnt[j] = c; cap += c;
    }
    csc_alloc(h, n, n, cap);
    for (int i = 0; i < n; ++i) mark[i] = -1;
    int pos = 0;
    for (int j = 0; j < n; ++j) {
        int c = 0;
        h->colptr[j] = pos;
        /* (A^T A)(:,j) = sum_k At(:,k) * A(k,j), k increasing */
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p) {
            int k = a->rowidx[p];
            double v = a->val[p];
            for (int s = at->colptr[k]; s < at->colptr[k + 1]; ++s) {
                int i = at->rowidx[s];
                if (mark[i] != j) { mark[i] = j; list[c++] = i; acc[i] = at->val[s] * v; }
                else acc[i] += at->val[s] * v;
            }
        }
        if (mark[j] != j) { mark[j] = j; list[c++] = j; acc[j] = 0.0; }
        /* sort row indices */
        for (int u = 1; u < c; ++u) {
            int key = list[u], w = u - 1;
            while (w >= 0 && list[w] > key) { list[w + 1] = list[w]; --w; }
            list[w + 1] = key;
        }
        for (int u = 0; u < c; ++u) {
            int i = list[u];
            double e = rho * acc[i];
            if (i == j) e = Qd[j] + e;
            h->rowidx[pos] = i;
            h->val[pos] = 2.0 * e;
            ++pos;
        }
    }
    h->colptr[n] = pos;
    h->nnz = pos;
    free(acc); free(mark); free(list); free(cnt);
}

/* ------------------------------------------------------------ ProblemData --- */
typedef struct {
    int n, mrows;
    csc_t A, At, H;
    int have_mats;
    const double *Qd, *q, *lb, *ub;
    double *bPk, *ATbPk;
    double *x_k, *x_k_1, *y_k, *y_k_1, *y_diff, *grad, *r1, *r0;
    double rho, G_k_norm;
} prob_t;

static void prob_init(prob_t *p, int n, int mrows) {
    memset(p, 0, sizeof(*p));
    p->n = n; p->mrows = mrows;
    p->bPk = (double *)calloc((size_t)mrows, sizeof(double));
    p->r1 = (double *)calloc((size_t)mrows, sizeof(double));
    p->r0 = (double *)calloc((size_t)mrows, sizeof(double));
    p->ATbPk = (double *)calloc((size_t)n, sizeof(double));
    p->x_k = (double *)calloc((size_t)n, sizeof(double));
    p->x_k_1 = (double *)calloc((size_t)n, sizeof(double));
    p->y_k = (double *)calloc((size_t)n, sizeof(double));
    p->y_k_1 = (double *)calloc((size_t)n, sizeof(double));
    p->y_diff = (double *)calloc((size_t)n, sizeof(double));
    p->grad = (double *)calloc((size_t)n, sizeof(double));
}

static void prob_drop_mats(prob_t *p) {
    if (p->have_mats) { csc_free(&p->A); csc_free(&p->At); csc_free(&p->H); p->have_mats = 0; }
}

static void prob_free(prob_t *p) {
    prob_drop_mats(p);
    free(p->bPk); free(p->r1); free(p->r0); free(p->ATbPk);
    free(p->x_k); free(p->x_k_1); free(p->y_k); free(p->y_k_1); free(p->y_diff); free(p->grad);
}

/* problem.cpp:31-39.  Takes ownership of *A (the reference copies it by value). */
static void prob_set_data(prob_t *p, csc_t *A, const double *b, const double *P_k, double rho) {
    prob_drop_mats(p);
    p->A = *A; memset(A, 0, sizeof(*A));
    csc_transpose(&p->A, &p->At);
    p->rho = rho;
    build_hessian(&p->A, &p->At, p->Qd, rho, &p->H);
    p->have_mats = 1;
    for (int i = 0; i < p->mrows; ++i) p->bPk[i] = -b[i] + P_k[i];
    csc_tmatvec(&p->A, p->bPk, p->ATbPk);
    const double alpha = 2.0 * rho;
    for (int j = 0; j < p->n; ++j) p->ATbPk[j] = alpha * p->ATbPk[j] + (p->q ? p->q[j] : 0.0);
}

/* problem.cpp:54-56 */
static void prob_grad(prob_t *p) {
    csc_matvec(&p->H, p->y_k, p->grad);
    for (int j = 0; j < p->n; ++j) p->grad[j] += p->ATbPk[j];
}

/* problem.cpp:46-51 */
static double prob_obj_diff(prob_t *p) {
    double t1 = 0.0, t2 = 0.0, n1 = 0.0, n0 = 0.0;
    for (int j = 0; j < p->n; ++j) {
        double d = p->y_k_1[j] - p->y_k[j];
        t1 += ((p->y_k_1[j] + p->y_k[j]) * p->Qd[j]) * d;
        if (p->q) t2 += p->q[j] * d;
    }
    csc_matvec(&p->A, p->y_k_1, p->r1);
    csc_matvec(&p->A, p->y_k, p->r0);
    for (int i = 0; i < p->mrows; ++i) {
        double a = p->r1[i] + p->bPk[i], c = p->r0[i] + p->bPk[i];
        n1 += a * a; n0 += c * c;
    }
    return t1 + t2 + p->rho * (n1 - n0);
}

/* ------------------------------------------------------------------ FISTA --- */
typedef struct { double L, beta, mu; int soc; int n_bt; } fista_t;

/* fista.cpp:52-70 ("SoC" projection exactly as written there) */
static void soc_projection(prob_t *p, const fista_t *f) {
    const double mu = f->mu;
    for (int j = 0; j < p->n; ++j) p->y_k_1[j] = p->y_k[j] - p->grad[j] / f->L;
    for (int i = 0; i + 2 < p->n; i += 3) {
        double *y = p->y_k_1 + i;
        double s = y[0] * y[0] + y[1] * y[1];
        double z = y[2];
        if (s * mu < -z || z < 0) {
            y[0] = 0.0; y[1] = 0.0; y[2] = 0.0;
        } else if (s > mu * z) {
            double k = ((mu * mu) * s + (mu * z)) / (((mu * mu) + 1) * s);
            y[0] *= k; y[1] *= k;
            y[2] = (mu * s + z) / ((mu * mu) + 1);
        }
    }
}

/* fista.cpp:6-27 */
static void fista_step(prob_t *p, fista_t *f) {
    prob_grad(p);
    for (;;) {
        if (!f->soc) {
            for (int j = 0; j < p->n; ++j) {
                double v = p->y_k[j] - p->grad[j] / f->L;
                v = v < p->ub[j] ? v : p->ub[j];  /* cwiseMin(ub) */
                v = v > p->lb[j] ? v : p->lb[j];  /* cwiseMax(lb) */
                p->y_k_1[j] = v;
            }
        } else {
            soc_projection(p, f);
        }
        double g2 = 0.0, gd = 0.0;
        for (int j = 0; j < p->n; ++j) {
            p->y_diff[j] = p->y_k_1[j] - p->y_k[j];
            g2 += p->y_diff[j] * p->y_diff[j];
        }
        p->G_k_norm = sqrt(g2);
        for (int j = 0; j < p->n; ++j) gd += p->grad[j] * p->y_diff[j];
        double od = prob_obj_diff(p);
        if (od > gd + (f->L / 2) * (p->G_k_norm * p->G_k_norm)) {
            f->L = f->beta * f->L;
            f->n_bt++;
        } else {
            memcpy(p->x_k_1, p->y_k_1, sizeof(double) * (size_t)p->n);
            break;
        }
    }
}

/* fista.cpp:29-50; returns iterations executed */
static int fista_optimize(prob_t *p, fista_t *f, int max_iters, double tol) {
    memcpy(p->y_k, p->x_k, sizeof(double) * (size_t)p->n);
    double t_k = 1.0, t_k_1;
    int it = 0;
    for (int i = 0; i < max_iters; ++i) {
        fista_step(p, f);
        ++it;
        t_k_1 = 1.0 + sqrt(1 + 4 * t_k * t_k) / 2.0;  /* sic: fista.cpp:34 */
        const double c = (t_k - 1) / t_k_1;
        for (int j = 0; j < p->n; ++j)
            p->y_k_1[j] = p->x_k_1[j] + c * (p->x_k_1[j] - p->x_k[j]);
        memcpy(p->x_k, p->x_k_1, sizeof(double) * (size_t)p->n);
        if (p->G_k_norm < tol) break;
        memcpy(p->y_k, p->y_k_1, sizeof(double) * (size_t)p->n);
        t_k = t_k_1;
    }
    return it;
}

/* ------------------------------------------------------ CentroidalDynamics --- */
#define CNT(t, n) cnt_plan[(((t) * n_eff) + (n)) * 4 + 0]
#define RR(t, n, k) cnt_plan[(((t) * n_eff) + (n)) * 4 + 1 + (k)]

/* centroidal.cpp:57-84 */
static void compute_x_mat(int n_col, int n_eff, double m, const double *cnt_plan,
                          const double *dt, const double *X, csc_t *A_x, double *b_x) {
    const int nt = 9 * n_eff * n_col;
    int *tr = (int *)malloc(sizeof(int) * (size_t)nt), *tc = (int *)malloc(sizeof(int) * (size_t)nt);
    double *tv = (double *)malloc(sizeof(double) * (size_t)nt);
    int k = 0;
    for (int i = 0; i < 9 * (n_col + 1); ++i) b_x[i] = 0.0;
    for (int t = 0; t < n_col; ++t) {
        b_x[9 * t + 3] = X[9 * (t + 1) + 3] - X[9 * t + 3];
        b_x[9 * t + 4] = X[9 * (t + 1) + 4] - X[9 * t + 4];
        b_x[9 * t + 5] = X[9 * (t + 1) + 5] - X[9 * t + 5] + 9.81 * dt[t];
        b_x[9 * t + 6] = X[9 * (t + 1) + 6] - X[9 * t + 6];
        b_x[9 * t + 7] = X[9 * (t + 1) + 7] - X[9 * t + 7];
        b_x[9 * t + 8] = X[9 * (t + 1) + 8] - X[9 * t + 8];
        for (int n = 0; n < n_eff; ++n) {
            const int c0 = 3 * n_eff * t + 3 * n;
            const double c = CNT(t, n);
#define PUT(r_, c_, v_) do { tr[k] = (r_); tc[k] = (c_); tv[k] = (v_); ++k; } while (0)
            PUT(9 * t + 3, c0 + 0, c * (dt[t] / m));
            PUT(9 * t + 4, c0 + 1, c * (dt[t] / m));
            PUT(9 * t + 5, c0 + 2, c * (dt[t] / m));
            PUT(9 * t + 6, c0 + 1, c * (X[9 * t + 2] - RR(t, n, 2)) * dt[t]);
            PUT(9 * t + 6, c0 + 2, -c * (X[9 * t + 1] - RR(t, n, 1)) * dt[t]);
            PUT(9 * t + 7, c0 + 0, -c * (X[9 * t + 2] - RR(t, n, 2)) * dt[t]);
            PUT(9 * t + 7, c0 + 2, c * (X[9 * t + 0] - RR(t, n, 0)) * dt[t]);
            PUT(9 * t + 8, c0 + 0, c * (X[9 * t + 1] - RR(t, n, 1)) * dt[t]);
            PUT(9 * t + 8, c0 + 1, -c * (X[9 * t + 0] - RR(t, n, 0)) * dt[t]);
        }
    }
    csc_from_triplets(A_x, 9 * (n_col + 1), 3 * n_eff * n_col, k, tr, tc, tv);
    free(tr); free(tc); free(tv);
}

/* centroidal.cpp:6-37 (static pattern), 86-127 (values), centroidal.hpp:22-27 (x_init rows) */
static void compute_f_mat(int n_col, int n_eff, double m, const double *cnt_plan,
                          const double *dt, const double *F, const double *x_init,
                          csc_t *A_f, double *b_f) {
    const int cap = n_col * (18 + 3 + 6) + 9;
    int *tr = (int *)malloc(sizeof(int) * (size_t)cap), *tc = (int *)malloc(sizeof(int) * (size_t)cap);
    double *tv = (double *)malloc(sizeof(double) * (size_t)cap);
    int k = 0;
    for (int i = 0; i < 9 * (n_col + 1); ++i) b_f[i] = 0.0;
    for (int t = 0; t < n_col; ++t) {
        for (int l = 0; l < 9; ++l) {
            PUT(9 * t + l, 9 * t + l, 1.0);
            PUT(9 * t + l, 9 * (t + 1) + l, -1.0);
        }
        for (int l = 0; l < 3; ++l) PUT(9 * t + l, 9 * (t + 1) + 3 + l, dt[t]);
        double a61 = 0, a62 = 0, a70 = 0, a72 = 0, a80 = 0, a81 = 0;
        double b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0;
        for (int n = 0; n < n_eff; ++n) {
            const double c = CNT(t, n);
            const double fx = F[3 * t * n_eff + 3 * n + 0], fy = F[3 * t * n_eff + 3 * n + 1],
                         fz = F[3 * t * n_eff + 3 * n + 2];
            const double rx = RR(t, n, 0), ry = RR(t, n, 1), rz = RR(t, n, 2);
            const double e61 = -c * fz * dt[t], e62 = c * fy * dt[t];
            const double e70 = c * fz * dt[t], e72 = -c * fx * dt[t];
            const double e80 = -c * fy * dt[t], e81 = c * fx * dt[t];
            const double d3 = -c * fx * dt[t] / m, d4 = -c * fy * dt[t] / m, d5 = -c * fz * dt[t] / m;
            const double d6 = (c * fy * rz - c * fz * ry) * dt[t];
            const double d7 = (c * fz * rx - c * fx * rz) * dt[t];
            const double d8 = (c * fx * ry - c * fy * rx) * dt[t];
            if (n == 0) {
                a61 = e61; a62 = e62; a70 = e70; a72 = e72; a80 = e80; a81 = e81;
                b3 = d3; b4 = d4; b5 = d5 + 9.81 * dt[t]; b6 = d6; b7 = d7; b8 = d8;
            } else {
                a61 += e61; a62 += e62; a70 += e70; a72 += e72; a80 += e80; a81 += e81;
                b3 += d3; b4 += d4; b5 += d5; b6 += d6; b7 += d7; b8 += d8;
            }
        }
        PUT(9 * t + 6, 9 * t + 1, a61); PUT(9 * t + 6, 9 * t + 2, a62);
        PUT(9 * t + 7, 9 * t + 0, a70); PUT(9 * t + 7, 9 * t + 2, a72);
        PUT(9 * t + 8, 9 * t + 0, a80); PUT(9 * t + 8, 9 * t + 1, a81);
        b_f[9 * t + 3] = b3; b_f[9 * t + 4] = b4; b_f[9 * t + 5] = b5;
        b_f[9 * t + 6] = b6; b_f[9 * t + 7] = b7; b_f[9 * t + 8] = b8;
    }
    for (int l = 0; l < 9; ++l) {
        PUT(9 * n_col + l, l, 1.0);
        b_f[9 * n_col + l] = x_init[l];
    }
    csc_from_triplets(A_f, 9 * (n_col + 1), 9 * (n_col + 1), k, tr, tc, tv);
    free(tr); free(tc); free(tv);
}
#undef PUT

/* ------------------------------------------------------------- BiConvexMP --- */
void orc_default_params(orc_params_t *p) {
    p->rho = 1e5; p->beta = 1.5; p->mu = 1.0; p->tol = 1e-5; p->exit_tol = 1e-3; p->maxit = 150;
}

static int orc_biconvex_solve_traced(int n_col, int n_eff, double m, const orc_params_t *prm,
                              const double *cnt_plan, const double *dt, const double *x_init,
                              const double *Qx, const double *qx,
                              const double *Qf, const double *qf,
                              const double *lbx, const double *ubx,
                              double *X, double *F, double *P,
                              double *L_x, double *L_f, int num_iters,
                              double *dyn_viol_hist, int *stats, int *trace) {
    const int nx = 9 * (n_col + 1), nf = 3 * n_eff * n_col;
    prob_t px, pf; /* prob_data_x holds X (uses A_f); prob_data_f holds F (uses A_x) */
    prob_init(&px, nx, nx);
    prob_init(&pf, nf, nx);
    px.Qd = Qx; px.q = qx; px.lb = lbx; px.ub = ubx;
    pf.Qd = Qf; pf.q = qf; pf.lb = NULL; pf.ub = NULL;
    memcpy(px.x_k, X, sizeof(double) * (size_t)nx);
    memcpy(pf.x_k, F, sizeof(double) * (size_t)nf);
    fista_t fx = {*L_x, prm->beta, prm->mu, 0, 0};
    fista_t ff = {*L_f, prm->beta, prm->mu, 1, 0}; /* biconvex.cpp:24 set_soc_true */
    double *b = (double *)malloc(sizeof(double) * (size_t)nx);
    double *viol = (double *)malloc(sizeof(double) * (size_t)nx);
    int status = 0, it_admm = 0, it_f = 0, it_x = 0;

    for (int i = 0; i < num_iters; ++i) {
        csc_t A;
        /* optimizing for F: biconvex.cpp:89-91 */
        compute_x_mat(n_col, n_eff, m, cnt_plan, dt, px.x_k, &A, b);
        prob_set_data(&pf, &A, b, P, prm->rho);
        it_f += fista_optimize(&pf, &ff, prm->maxit, prm->tol);
        /* optimizing for X: biconvex.cpp:94-96 */
        compute_f_mat(n_col, n_eff, m, cnt_plan, dt, pf.x_k, x_init, &A, b);
        prob_set_data(&px, &A, b, P, prm->rho);
        it_x += fista_optimize(&px, &fx, prm->maxit, prm->tol);
        /* biconvex.cpp:98-99 */
        csc_matvec(&px.A, px.x_k, viol);
        double nrm2 = 0.0;
        for (int r = 0; r < nx; ++r) { viol[r] -= b[r]; P[r] += viol[r]; nrm2 += viol[r] * viol[r]; }
        const double nrm = sqrt(nrm2);
        ++it_admm;
        if (dyn_viol_hist) dyn_viol_hist[i] = nrm;
        if (trace) { trace[4 * i] = it_f; trace[4 * i + 1] = it_x; trace[4 * i + 2] = ff.n_bt; trace[4 * i + 3] = fx.n_bt; }
        if (isnan(nrm)) { status = 2; break; }   /* biconvex.cpp:106-109 */
        if (nrm < prm->exit_tol) break;          /* biconvex.cpp:111-114 */
    }
    memcpy(X, px.x_k, sizeof(double) * (size_t)nx);
    memcpy(F, pf.x_k, sizeof(double) * (size_t)nf);
    *L_x = fx.L; *L_f = ff.L;
    if (stats) {
        stats[0] = it_admm; stats[1] = it_f; stats[2] = it_x;
        stats[3] = ff.n_bt; stats[4] = fx.n_bt; stats[5] = status;
    }
    free(b); free(viol);
    prob_free(&px); prob_free(&pf);
    return status;
}

int orc_biconvex_solve(int n_col, int n_eff, double m, const orc_params_t *prm,
                       const double *cnt_plan, const double *dt, const double *x_init,
                       const double *Qx, const double *qx,
                       const double *Qf, const double *qf,
                       const double *lbx, const double *ubx,
                       double *X, double *F, double *P,
                       double *L_x, double *L_f, int num_iters,
                       double *dyn_viol_hist, int *stats) {
    return orc_biconvex_solve_traced(n_col, n_eff, m, prm, cnt_plan, dt, x_init, Qx, qx, Qf, qf, lbx, ubx, X, F, P, L_x, L_f,
                                     num_iters, dyn_viol_hist, stats, NULL);
}

int orc_biconvex_solve_batch_traced(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                                    const double *cnt_plan, const double *dt, const double *x_init,
                                    const double *Qx, const double *qx,
                                    const double *Qf, const double *qf,
                                    const double *lbx, const double *ubx, int shared_cost,
                                    double *X, double *F, double *P,
                                    double *L_x, double *L_f, int num_iters,
                                    int *stats, int nthreads, double *hist, int *trace) {
    const int nx = 9 * (n_col + 1), nf = 3 * n_eff * n_col;
    int ndiv = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : ndiv)
    for (int b = 0; b < B; ++b) {
        const size_t cb = shared_cost ? 0 : (size_t)b;
        int st = orc_biconvex_solve_traced(
            n_col, n_eff, m, prm, cnt_plan + (size_t)b * n_col * n_eff * 4, dt + (size_t)b * n_col,
            x_init + (size_t)b * 9, Qx + cb * nx, qx + (size_t)b * nx, Qf + cb * nf,
            qf ? qf + cb * nf : NULL, lbx + (size_t)b * nx, ubx + (size_t)b * nx,
            X + (size_t)b * nx, F + (size_t)b * nf, P + (size_t)b * nx, L_x + b, L_f + b,
            num_iters, hist ? hist + (size_t)b * num_iters : NULL, stats ? stats + (size_t)b * ORC_NSTATS : NULL,
            trace ? trace + (size_t)b * num_iters * 4 : NULL);
        ndiv += (st != 0);
    }
    return ndiv;
}

int orc_biconvex_solve_batch(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                             const double *cnt_plan, const double *dt, const double *x_init,
                             const double *Qx, const double *qx,
                             const double *Qf, const double *qf,
                             const double *lbx, const double *ubx, int shared_cost,
                             double *X, double *F, double *P,
                             double *L_x, double *L_f, int num_iters,
                             int *stats, int nthreads) {
    return orc_biconvex_solve_batch_traced(B, n_col, n_eff, m, prm, cnt_plan, dt, x_init, Qx, qx, Qf, qf, lbx, ubx, shared_cost,
                                           X, F, P, L_x, L_f, num_iters, stats, nthreads, NULL, NULL);
}

/* biconvex.cpp:27-55 */
void orc_create_bound_constraints(int n_col, int n_eff, const double *cnt_plan,
                                  const double *b, double *lbx, double *ubx) {
    const int nx = 9 * (n_col + 1);
    for (int i = 0; i < nx; ++i) { lbx[i] = -INFINITY; ubx[i] = INFINITY; }
    for (int i = 0; i < n_col; ++i) {
        double csum = 0.0;
        for (int n = 0; n < n_eff; ++n) csum += CNT(i, n);
        if (csum > 0) {
            for (int k = 0; k < 3; ++k) {
                double mx = RR(i, 0, k), mn = RR(i, 0, k);
                for (int n = 1; n < n_eff; ++n) {
                    if (RR(i, n, k) > mx) mx = RR(i, n, k);
                    if (RR(i, n, k) < mn) mn = RR(i, n, k);
                }
                lbx[9 * i + k] = mx + b[6 * i + k];
                ubx[9 * i + k] = mn + b[6 * i + 3 + k];
            }
        }
    }
}

/* biconvex.cpp:57-72 */
void orc_create_cost_X(int n_col, const double *W_X, const double *W_X_ter,
                       const double *X_ter, const double *X_nom, double *Qx, double *qx) {
    const int nv = 9 * (n_col + 1);
    for (int i = 0; i < nv - 9; ++i) { Qx[i] = W_X[i]; qx[i] = -2 * (X_nom[i] * W_X[i]); }
    for (int i = nv - 9; i < nv; ++i) {
        Qx[i] = W_X_ter[i - nv + 9];
        qx[i] = -2 * (X_ter[i - nv + 9] * W_X_ter[i - nv + 9]);
    }
}

static void csc_to_dense(const csc_t *a, double *d) {
    memset(d, 0, sizeof(double) * (size_t)a->rows * (size_t)a->cols);
    for (int j = 0; j < a->cols; ++j)
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p)
            d[(size_t)a->rowidx[p] * a->cols + j] = a->val[p];
}

void orc_dense_A_x(int n_col, int n_eff, double m, const double *cnt_plan, const double *dt,
                   const double *X, double *A_x, double *b_x) {
    csc_t A;
    compute_x_mat(n_col, n_eff, m, cnt_plan, dt, X, &A, b_x);
    csc_to_dense(&A, A_x);
    csc_free(&A);
}

void orc_dense_A_f(int n_col, int n_eff, double m, const double *cnt_plan, const double *dt,
                   const double *F, const double *x_init, double *A_f, double *b_f) {
    csc_t A;
    compute_f_mat(n_col, n_eff, m, cnt_plan, dt, F, x_init, &A, b_f);
    csc_to_dense(&A, A_f);
    csc_free(&A);
}

/* ------------------------------------------------------------ QuadrupedGait --- */
/* gait_planner.cpp:41-44 */
double orc_gait_phi(double t, double gait_period, double phase_offset) {
    return fmod(t + phase_offset * gait_period, gait_period);
}

/* gait_planner.cpp:46-58; the unqualified abs() there binds to the double overload
 * when <Eigen/Dense> is included under libstdc++ (stdlib.h wrapper), restated as fabs */
int orc_gait_phase(double t, double gait_period, double stance_percent, double phase_offset) {
    const double stance_time = gait_period * stance_percent;
    const double phi = orc_gait_phi(t, gait_period, phase_offset);
    return (phi <= stance_time || fabs(phi - stance_time) < 1e-4) ? 1 : 0;
}

/* gait_planner.cpp:112-128 */
double orc_gait_percent_in_phase(double t, double gait_period, double stance_percent,
                                 double phase_offset) {
    const double stance_time = gait_period * stance_percent;
    const double phi = orc_gait_phi(t, gait_period, phase_offset);
    if (phi <= stance_time) return phi / stance_time;
    return (phi - stance_time) / (gait_period - stance_time);
}

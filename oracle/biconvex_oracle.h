/*
 * oracle/biconvex_oracle.h -- CPU restatement of the reference's centroidal
 * bi-convex ADMM solve (BiConvexMP::optimize and what it calls).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, link, import or call it, and only as the checker / the timed CPU
 * baseline.  The product (bunmpc_amd/) never includes or links this file.
 *
 * PARITY UNPINNED: the reference holds no golden vectors / known-answer tests
 * for this path and cannot be compiled or imported
 * in this image (Eigen 3 / pinocchio / crocoddyl are absent, no network).
 * This restatement is therefore pinned only by (1) a second, independent
 * numpy/scipy.sparse restatement (oracle/oracle_np.py) that must agree with
 * it to ~1e-12 and (2) analytic invariants checked in tests/.
 *
 * Reference files followed (paths relative to
 * /root/reference/iterative_supervised_learning/):
 *   src/motion_planner/biconvex.cpp:6-25,27-78,80-120   (ctor, costs, ADMM)
 *   include/motion_planner/biconvex.hpp:146-160          (defaults)
 *   src/dynamics/centroidal.cpp:6-37,39-49,57-84,86-127  (A_x,b_x,A_f,b_f)
 *   include/dynamics/centroidal.hpp:22-27                (update_x_init)
 *   src/solvers/problem.cpp:31-56                        (set_data, grad, obj diff)
 *   src/solvers/fista.cpp:6-27,29-50,52-70               (FISTA, "SoC" projection)
 *   include/solvers/fista.hpp:49-60                      (L_, beta_, mu_)
 */
#ifndef BUNMPC_BICONVEX_ORACLE_H
#define BUNMPC_BICONVEX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Solver constants as the reference hard-codes them. */
typedef struct {
    double rho;       /* biconvex.hpp:148 default 1e5; harness set_rho(5e4)      */
    double beta;      /* fista.hpp:53  1.5                                        */
    double mu;        /* fista.hpp:60  1.0                                        */
    double tol;       /* biconvex.hpp:158 1e-5                                    */
    double exit_tol;  /* biconvex.hpp:160 1e-3                                    */
    int    maxit;     /* biconvex.hpp:156 150                                     */
} orc_params_t;

void orc_default_params(orc_params_t *p);

/* stats layout written by orc_biconvex_solve (ints):
 * [0] ADMM iterations executed   [1] sum of F-step FISTA iterations
 * [2] sum of X-step FISTA iters  [3] F-step backtracking retries
 * [4] X-step backtracking retries [5] status: 0 ok, 2 NaN divergence      */
#define ORC_NSTATS 6

/*
 * One BiConvexMP::optimize(x_init, num_iters) (biconvex.cpp:80-120).
 *   cnt_plan : [n_col][n_eff][4] rows [flag,x,y,z] (set_contact_plan, centroidal.cpp:39-49)
 *   dt       : [n_col]
 *   Qx, qx   : diagonal of Q_ and q_ of the X problem, length 9(n_col+1)
 *   Qf, qf   : same for the F problem, length 3*n_eff*n_col (qf may be NULL = 0)
 *   lbx, ubx : box bounds of the X problem, length 9(n_col+1)
 *   X, F, P  : in = warm start (set_warm_start_vars), out = last iterates
 *   L_x, L_f : FISTA step constants, persistent across calls (fista.hpp:52)
 *   dyn_viol_hist : num_iters doubles or NULL (collect_statistics)
 * returns status (0 ok, 2 diverged/NaN).
 */
int orc_biconvex_solve(int n_col, int n_eff, double m, const orc_params_t *prm,
                       const double *cnt_plan, const double *dt, const double *x_init,
                       const double *Qx, const double *qx,
                       const double *Qf, const double *qf,
                       const double *lbx, const double *ubx,
                       double *X, double *F, double *P,
                       double *L_x, double *L_f, int num_iters,
                       double *dyn_viol_hist, int *stats);

/* Batch of independent solves (fresh L0 per problem is the caller's job);
 * arrays are the single-problem arrays stacked along a leading batch axis.
 * weights_shared != 0: Qx,qx? no -- only Qx, Qf, lbx/ubx offsets may be shared:
 *   shared_cost   : Qx, Qf (and qf) are one copy for the whole batch
 * nthreads <= 0 -> all OpenMP threads.  Returns number of diverged problems. */
int orc_biconvex_solve_batch(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                             const double *cnt_plan, const double *dt, const double *x_init,
                             const double *Qx, const double *qx,
                             const double *Qf, const double *qf,
                             const double *lbx, const double *ubx, int shared_cost,
                             double *X, double *F, double *P,
                             double *L_x, double *L_f, int num_iters,
                             int *stats, int nthreads);

/* ... with the solve's history: hist [B][num_iters] = ||A_f X - b_f|| after every ADMM iteration (collect_statistics,
 * biconvex.cpp:101-104), trace [B][num_iters][4] = running totals {F-step FISTA iterations, X-step FISTA iterations, F-step
 * retries, X-step retries} after every ADMM iteration.  Rows of iterations that did not run are left untouched.  Either may
 * be NULL.  Used by the prefix-parity tests (tests/util.py::prefix_parity). */
int orc_biconvex_solve_batch_traced(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                                    const double *cnt_plan, const double *dt, const double *x_init,
                                    const double *Qx, const double *qx,
                                    const double *Qf, const double *qf,
                                    const double *lbx, const double *ubx, int shared_cost,
                                    double *X, double *F, double *P,
                                    double *L_x, double *L_f, int num_iters,
                                    int *stats, int nthreads, double *hist, int *trace);
int orc_fast_solve_batch_traced(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                                const double *cnt_plan, const double *dt, const double *x_init,
                                const double *Qx, const double *qx,
                                const double *Qf, const double *qf,
                                const double *lbx, const double *ubx, int shared_cost,
                                double *X, double *F, double *P,
                                double *L_x, double *L_f, int num_iters,
                                int *stats, int nthreads, double *hist, int *trace);

/* Same arguments and results, matrix-free (biconvex_fast.c): the faster CPU baseline of bench.py. */
int orc_fast_solve_batch(int B, int n_col, int n_eff, double m, const orc_params_t *prm,
                         const double *cnt_plan, const double *dt, const double *x_init,
                         const double *Qx, const double *qx,
                         const double *Qf, const double *qf,
                         const double *lbx, const double *ubx, int shared_cost,
                         double *X, double *F, double *P,
                         double *L_x, double *L_f, int num_iters,
                         int *stats, int nthreads);

/* biconvex.cpp:27-55 -- X box from the contact plan; b is [n_col][6]. */
void orc_create_bound_constraints(int n_col, int n_eff, const double *cnt_plan,
                                  const double *b, double *lbx, double *ubx);
/* biconvex.cpp:57-72 -- W_X[9 n_col], W_X_ter[9], X_ter[9], X_nom[9 n_col]. */
void orc_create_cost_X(int n_col, const double *W_X, const double *W_X_ter,
                       const double *X_ter, const double *X_nom, double *Qx, double *qx);

/* Dense copies of the matrices for tests (return_A_x / return_b_x / return_A_f /
 * return_b_f, biconvex.hpp:30-53).  A_x: [9(H+1)][3EH] row-major; A_f: [9(H+1)]^2. */
void orc_dense_A_x(int n_col, int n_eff, double m, const double *cnt_plan, const double *dt,
                   const double *X, double *A_x, double *b_x);
void orc_dense_A_f(int n_col, int n_eff, double m, const double *cnt_plan, const double *dt,
                   const double *F, const double *x_init, double *A_f, double *b_f);

/* gait_planner.cpp:41-58,112-128 (scalar overloads). */
double orc_gait_phi(double t, double gait_period, double phase_offset);
int    orc_gait_phase(double t, double gait_period, double stance_percent, double phase_offset);
double orc_gait_percent_in_phase(double t, double gait_period, double stance_percent,
                                 double phase_offset);

#ifdef __cplusplus
}
#endif
#endif

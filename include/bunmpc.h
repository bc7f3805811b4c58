/*
 * include/bunmpc.h -- C-ABI of libbunmpc_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the BiConvex MPC solve path of Atarilab/BUNMPC.  The
 * reference crosses Python -> C++ through three pybind11 modules
 * (iterative_supervised_learning/srcpy/...); each entry point below names the
 * binding line it replaces.  Plain pointers and sizes only; every function that
 * can fail returns an int status and never throws:
 *      BMPC_OK 0, BMPC_BAD_ARG 1, BMPC_DIVERGED 2 (NaN in the dynamics violation),
 *      BMPC_DEVICE_ERROR 3 (HIP error; text via bmpc_last_error()).
 * Matrices are row-major doubles unless stated.  The single-problem handles keep
 * the reference's call semantics (append-style set_contact_plan, persistent FISTA
 * step constants, state surviving between optimize calls); the *_batch entry
 * points are additive and solve many independent problems in one kernel launch.
 * All compute runs on the GPU: there is no CPU fallback in this library.
 */
#ifndef BUNMPC_H
#define BUNMPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define BMPC_OK 0
#define BMPC_BAD_ARG 1
#define BMPC_DIVERGED 2
#define BMPC_DEVICE_ERROR 3

/* library / device ----------------------------------------------------------- */
int bmpc_abi_version(void);
int bmpc_batch_struct_size(void);            /* sizeof(bmpc_batch_t), to catch binding drift  */
const char *bmpc_last_error(void);           /* thread-local text of the last failure   */
int bmpc_device_count(int *count);           /* hipGetDeviceCount                        */
int bmpc_set_device(int device);             /* hipSetDevice for the calling thread      */
/* DPP / permlane self test of the lane exchanges the kernels rely on; 0 = pass.     */
int bmpc_selftest_lanes(void);

/* gait_planner_cpp.GaitPlanner --------------------------------------------------
 * srcpy/gait_planner/py_gait_planner.cpp:19-35 over src/gait_planner/gait_planner.cpp */
typedef struct bmpc_gait bmpc_gait_t;
bmpc_gait_t *bmpc_gait_create(double gait_period, const double *stance_percent,
                              const double *phase_offset, int n_eff, double step_height); /* :22 */
void bmpc_gait_destroy(bmpc_gait_t *g);
int bmpc_gait_n_eff(const bmpc_gait_t *g);
int bmpc_gait_get_phase(bmpc_gait_t *g, double t, int foot_id, int *phase);             /* :25 */
int bmpc_gait_get_phase_all(bmpc_gait_t *g, double t, int *phase);                      /* :26 */
int bmpc_gait_get_phi(bmpc_gait_t *g, double t, int foot_id, double *phi);              /* :27 */
int bmpc_gait_get_phi_all(bmpc_gait_t *g, double t, double *phi);                       /* :28 */
int bmpc_gait_get_percent_in_phase(bmpc_gait_t *g, double t, int foot_id, double *pct); /* :29-30 */
int bmpc_gait_get_percent_in_phase_all(bmpc_gait_t *g, double t, double *pct);          /* :31-32 */
/* plan: rows x n_eff ints, row i = get_phase_all(t + i*dt)                             :33 */
int bmpc_gait_get_contact_phase_plan(bmpc_gait_t *g, int rows, double t, double dt, int *plan);
int bmpc_gait_set_step_height(bmpc_gait_t *g, double step_height);                      /* :34 */
int bmpc_gait_set_stance_percent(bmpc_gait_t *g, double lf, double lh, double rf, double rh); /* :35 */

/* biconvex_mpc_cpp.BiconvexMP ----------------------------------------------------
 * srcpy/motion_planner/biconvex.cpp:19-44 over src/motion_planner/biconvex.cpp */
typedef struct bmpc_biconvex bmpc_biconvex_t;
bmpc_biconvex_t *bmpc_biconvex_create(double m, int n_col, int n_eff);                  /* :20 */
void bmpc_biconvex_destroy(bmpc_biconvex_t *h);
int bmpc_biconvex_n_col(const bmpc_biconvex_t *h);
int bmpc_biconvex_n_eff(const bmpc_biconvex_t *h);
/* cnt_plan: n_eff x 4 rows [flag,x,y,z]; appended (H calls per solve)                  :21 */
int bmpc_biconvex_set_contact_plan(bmpc_biconvex_t *h, const double *cnt_plan, double dt);
int bmpc_biconvex_set_rotation_matrix_f(bmpc_biconvex_t *h, const double *R3x3);        /* :22 */
/* dense debugging copies; A_x: 9(H+1) x 3EH, A_f: 9(H+1) x 9(H+1)                      :23-26 */
int bmpc_biconvex_return_A_x(bmpc_biconvex_t *h, const double *X, double *A_x);
int bmpc_biconvex_return_b_x(bmpc_biconvex_t *h, const double *X, double *b_x);
int bmpc_biconvex_return_A_f(bmpc_biconvex_t *h, const double *F, const double *x_init, double *A_f);
int bmpc_biconvex_return_b_f(bmpc_biconvex_t *h, const double *F, const double *x_init, double *b_f);
/* Q is passed as its diagonal (the shim rejects off-diagonal entries)                  :27,29 */
int bmpc_biconvex_set_cost_x(bmpc_biconvex_t *h, const double *Q_diag, const double *q);
int bmpc_biconvex_set_cost_f(bmpc_biconvex_t *h, const double *Q_diag, const double *q);
int bmpc_biconvex_create_cost_X(bmpc_biconvex_t *h, const double *W_X, const double *W_X_ter,
                                const double *X_ter, const double *X_nom);              /* :28 */
int bmpc_biconvex_create_cost_F(bmpc_biconvex_t *h, const double *W_F);                 /* :30 */
int bmpc_biconvex_set_bounds_x(bmpc_biconvex_t *h, const double *lb, const double *ub); /* :31 */
int bmpc_biconvex_set_bounds_f(bmpc_biconvex_t *h, const double *lb, const double *ub); /* :32 */
/* b: rows x cols (cols must be 6, rows >= n_col)                                       :33 */
int bmpc_biconvex_create_bound_constraints(bmpc_biconvex_t *h, const double *b, int rows, int cols,
                                           double fx_max, double fy_max, double fz_max);
int bmpc_biconvex_set_rho(bmpc_biconvex_t *h, double rho);                              /* :34 */
int bmpc_biconvex_return_opt_x(bmpc_biconvex_t *h, double *X);      /* 9(H+1)           :35 */
int bmpc_biconvex_return_opt_f(bmpc_biconvex_t *h, double *F);      /* 3EH              :36 */
int bmpc_biconvex_return_opt_p(bmpc_biconvex_t *h, double *P);      /* 9(H+1)           :37 */
int bmpc_biconvex_return_opt_com(bmpc_biconvex_t *h, double *com);  /* (H+1) x 3        :38 */
int bmpc_biconvex_return_opt_mom(bmpc_biconvex_t *h, double *mom);  /* (H+1) x 6        :39 */
int bmpc_biconvex_set_warm_start_vars(bmpc_biconvex_t *h, const double *X, const double *F,
                                      const double *P);                                 /* :41 */
/* returns BMPC_DIVERGED after writing NaNs, like the reference prints and returns      :42 */
int bmpc_biconvex_optimize(bmpc_biconvex_t *h, const double *x_init, int num_iters);
int bmpc_biconvex_dyn_viol_hist_size(const bmpc_biconvex_t *h);                         /* :43 */
int bmpc_biconvex_return_dyn_viol_hist(const bmpc_biconvex_t *h, double *hist);
int bmpc_biconvex_collect_statistics(bmpc_biconvex_t *h);                               /* :44 */
/* additive: FISTA step constants carried by the handle (fista.hpp:52), solver counters
 * of the last optimize: {admm iters, F-FISTA iters, X-FISTA iters, F retries, X retries, status} */
int bmpc_biconvex_get_step_constants(const bmpc_biconvex_t *h, double *L_x, double *L_f);
int bmpc_biconvex_set_step_constants(bmpc_biconvex_t *h, double L_x, double L_f);
int bmpc_biconvex_last_stats(const bmpc_biconvex_t *h, int *stats6);
/* not bound by the reference but public on the C++ class (biconvex.hpp:131-137) */
int bmpc_biconvex_set_friction_coefficient(bmpc_biconvex_t *h, double mu);
int bmpc_biconvex_set_robot_mass(bmpc_biconvex_t *h, double m);

/* batch of independent solves (additive) ------------------------------------------
 * One kernel launch = B x BiConvexMP::optimize.  Array shapes, batch outermost:
 *   cnt_plan [B][H][E][4], dt [B][H], x_init [B][9]
 *   harness form (raw = 0): W_X [.][9H], W_X_ter [.][9], W_F [.][3EH], bounds [.][H][6]
 *       with batch strides s* in doubles (0 = one copy shared by the batch),
 *       X_nom [B][9H], X_ter [B][9]      -- the kernel applies create_cost_X /
 *       create_cost_F / create_bound_constraints (biconvex.cpp:27-78) itself
 *   raw form (raw = 1): Qx, qx, lbx, ubx [B][9(H+1)], Qf [B][3EH], qf [B][3EH] or NULL
 *   X [B][9(H+1)], F [B][3EH], P [B][9(H+1)], L_x [B], L_f [B]   in: warm start, out: result
 *   dyn_viol [B] or NULL, hist [B][num_iters] or NULL, stats [B][6] or NULL
 */
typedef struct {
    int B, n_col, n_eff, raw;
    int num_iters, maxit;
    int cold_start;   /* 1: ignore X/F/P/L_x/L_f on entry and start as KinoDynMP::set_warm_starts does
                         (kino_dyn.cpp:83-99): X = tile(x_init), F = 0, P = 0, L = BMPC_L0_X / BMPC_L0_F */
    int reserved_;
    double m, rho, mu, beta, tol, exit_tol;
    const double *cnt_plan, *dt, *x_init;
    const double *W_X, *W_X_ter, *W_F, *bounds, *X_nom, *X_ter;
    long sW_X, sW_X_ter, sW_F, sbounds;
    const double *Qx, *qx, *lbx, *ubx, *Qf, *qf;
    double *X, *F, *P, *L_x, *L_f;
    double *dyn_viol, *hist;
    int *stats;
} bmpc_batch_t;

/* reference defaults: rho 1e5 (biconvex.hpp:148), mu 1, beta 1.5, tol 1e-5, exit_tol 1e-3,
 * maxit 150 (biconvex.hpp:152-160); L0 constants 2.25e6 / 506.25 (biconvex.cpp:20-21) */
void bmpc_batch_defaults(bmpc_batch_t *d);
#define BMPC_L0_X 2.25e6
#define BMPC_L0_F 506.25

/* all pointers are DEVICE pointers; asynchronous on hip_stream (a hipStream_t, NULL = default) */
int bmpc_biconvex_solve_batch_device(const bmpc_batch_t *d, void *hip_stream);
/* all pointers are HOST pointers; copies in, solves, copies out, synchronises */
int bmpc_biconvex_solve_batch_host(const bmpc_batch_t *d);
/* symbol-name prefix of the kernel that serves (n_col, raw), for profiles */
const char *bmpc_biconvex_kernel_name(int n_col, int raw);

#ifdef __cplusplus
}
#endif
#endif

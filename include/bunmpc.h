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
 *       with batch strides s* in doubles (0 = one copy shared by the batch; at most 2^26),
 *       X_nom [B][9H], X_ter [B][9]      -- the kernel applies create_cost_X /
 *       create_cost_F / create_bound_constraints (biconvex.cpp:27-78) itself
 *   raw form (raw = 1): Qx, qx, lbx, ubx [B][9(H+1)], Qf [B][3EH], qf [B][3EH] or NULL
 *   X [B][9(H+1)], F [B][3EH], P [B][9(H+1)], L_x [B], L_f [B]   in: warm start, out: result
 *   dyn_viol [B] or NULL, hist [B][num_iters] or NULL, stats [B][6] or NULL
 *   trace [B][num_iters][4] ints or NULL: running totals {F-step FISTA iterations, X-step FISTA iterations, F-step retries,
 *       X-step retries} after every ADMM iteration that ran (rows of iterations that did not run are not written) -- with hist
 *       the solve's discrete path per ADMM iteration, what the prefix-parity tests compare with the CPU oracle's
 *   A caller built against an older header must zero-initialise the whole struct (bmpc_batch_defaults does) and check
 *   bmpc_batch_struct_size() == sizeof(bmpc_batch_t).
 *   Shapes: n_eff = 4; n_col + 1 <= 256 knots (up to 64: 4 / 3 / 2 / 1 problems per wave, fp64 or fp32 iterates; 65 .. 256: one
 *   problem per workgroup of two .. four waves, fp64 -- the horizons of the reference's examples/analysis/solve_times_test.py).
 */
typedef struct {
    int B, n_col, n_eff, raw;
    int num_iters, maxit;
    int cold_start;   /* 0: X/F/P/L_x/L_f on entry are the warm start (set_warm_start_vars; L_ persists, fista.hpp:52).
                         1: a FRESH KinoDynMP per problem: ignore X/F/P/L_x/L_f on entry -- X = tile(x_init), F = 0, P = 0
                            (KinoDynMP::set_warm_starts, kino_dyn.cpp:83-99) AND L = BMPC_L0_X / BMPC_L0_F (the constructor's
                            values, biconvex.cpp:20-21): independent batch elements.
                         2: the NEXT optimize call of the same KinoDynMP objects: iterates reset as in 1, L_x / L_f taken from
                            the arrays (the reference never resets FISTA's L_ between calls): replans of the same rollouts. */
    int precision;    /* 0: fp64 arithmetic (reference behaviour).  1: fp32 iterates / operators / projections with
                         every accept / exit decision and the dynamics violation reduced in fp64; harness form
                         only; arrays stay fp64 in memory (BASELINE config 3) */
    double m, rho, mu, beta, tol, exit_tol;
    const double *cnt_plan, *dt, *x_init;
    const double *W_X, *W_X_ter, *W_F, *bounds, *X_nom, *X_ter;
    long sW_X, sW_X_ter, sW_F, sbounds;
    const double *Qx, *qx, *lbx, *ubx, *Qf, *qf;
    double *X, *F, *P, *L_x, *L_f;
    double *dyn_viol, *hist;
    int *stats;
    int *trace;       /* ABI version 2 */
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
/* Kernel selection (no effect on the discrete path; values equal to rounding): batches of at most max_batch problems with
 * n_col + 1 <= 21 knots in fp64 are solved ONE PROBLEM PER WAVE (knots x component groups across the lanes: the dependent
 * chain of a solve is ~2.3x shorter), larger ones one knot per lane with 4 / 2 / 1 problems per wave.  Default 1024 (one
 * wave per SIMD of an MI355X); 0 = never.  Returns the old value. */
int bmpc_set_latency_mapping_max_batch(int max_batch);
/* Horizons of 17..21 knots (n_col = 16..20: the headline shape) in fp64 can run THREE problems per wave, in 21-lane segments,
 * instead of two in 32-lane segments (same iterates; the segment sums behind the step decisions are added in another order, as
 * between any two of the mappings).  mode 0: never, 1: always, 2 (default): when it finishes the batch sooner -- fewer rounds of
 * waves over the chip's SIMDs (not at B = 4096 on an MI355X: 1366 waves or 2048, two rounds either way; at B = 3072 or 6144 it
 * saves a round), or num_iters >= 25 (iteration counts differ per problem).  Returns the old value. */
int bmpc_set_three_per_wave(int mode);
/* With many ADMM iterations (num_iters >= 25) the early exit (biconvex.cpp:111-114) makes the iteration counts differ per problem
 * (34 .. 100 at the reference's num_iters = 100).  The three-per-wave kernel then runs as a PERSISTENT grid of one wave per SIMD in
 * which a segment whose problem has finished stores its results and takes the next unsolved problem from a device counter
 * ("biconvex_admm_steal_kernel"): the batch takes the sum of the problems' iterations over the segments, not the per-wave maxima.
 * A problem's result does not depend on it.  on = 0: never (a test switch).  Default 1.  Returns the old value. */
int bmpc_set_work_stealing(int on);
/* waves of that persistent grid (an experiment switch; 0, the default: as many as the chip holds at once).  Returns the old value. */
int bmpc_set_steal_grid(int waves);
/* The fp64 batch kernel (16 / 32 / 64 lanes per problem) exists in two builds: for ONE wave per SIMD (FISTA iterates in registers,
 * ~290 of them) and for TWO (256 registers: x_k and its affine image rest in LDS between the iterations; same operations in the
 * same order, bit-identical results).  A lone wave of the second build is the slower one, two of them on a SIMD cover each
 * other's latencies.  mode 0: never the second, 1: always, 2 (default): when the batch needs more waves than the chip has SIMDs
 * (B = 4096 at n_col = 20 on an MI355X: 2048 waves over 1024 SIMDs).  Returns the old value. */
int bmpc_set_two_waves_per_simd(int mode);
/* lanes per problem of the calling host thread's latest batch solve: 16 / 21 / 32 / 64 (128 / 192 / 256: a workgroup of 2 / 3 / 4 waves), 0 = the one-problem-per-wave kernel */
int bmpc_biconvex_last_lanes_per_problem(void);
/* ... and the waves per SIMD its kernel was built for (1 or 2) */
int bmpc_biconvex_last_waves_per_simd(void);
/* The one-problem-per-wave kernel takes the two decisions of a FISTA step (retry, fista.cpp:16; exit, fista.cpp:29) from fp32
 * wave sums whenever both comparisons are clear of their thresholds by 1e-5 relative, from the fp64 sums and the reference
 * expression otherwise.  on = 1: always from the fp64 sums (a test switch: results must be bit-identical either way).
 * Returns the old value. */
int bmpc_set_exact_step_decisions(int on);
/* Scratch (private-segment) bytes per lane of the fp32 kernels as the loaded code object reports them, -1 on error.  0 is what
 * the build is set up for (bunmpc_amd/build.py: their translation unit is compiled without the SLP vectoriser); a compiler that
 * spills again shows up here, and in 4x the HBM traffic. */
int bmpc_biconvex_fp32_scratch_bytes(void);
/* symbol-name prefix of the kernel that serves (n_col, raw), for profiles */
const char *bmpc_biconvex_kernel_name(int n_col, int raw);
/* which kernel the calling host thread's latest batch solve was dispatched to: "biconvex_latency_kernel" (one problem per wave),
 * "biconvex_admm_kernel" or "biconvex_admm_kernel_f32" (one knot per lane); "" before the first solve */
const char *bmpc_biconvex_last_kernel_name(void);

/* rigid-body model -------------------------------------------------------------------
 * What pinocchio::urdf::buildModel(urdf, JointModelFreeFlyer()) yields (inverse_kinematics.cpp:10,
 * kino_dyn.cpp:9), as flat arrays (the Python shim parses the URDF: bunmpc_amd/urdf_model.py).
 * nj must be 12 (free-flyer + serial chains numbered contiguously, parents first).
 * R [nj][9], p [nj][3]: joint placement in the parent joint frame; axis [nj][3];
 * bodies 0..nj (0 = base): mass, com [.][3], inertia about the com [.][9] (joint frame);
 * frames: body index + position in that body's joint frame. */
typedef struct bmpc_model bmpc_model_t;
bmpc_model_t *bmpc_model_create(int nj, const int *parent, const double *R, const double *p, const double *axis,
                                const double *mass, const double *com, const double *inertia, int nframes,
                                const int *frame_body, const double *frame_p);
void bmpc_model_destroy(bmpc_model_t *m);
double bmpc_model_total_mass(const bmpc_model_t *m);

/* inverse_kinematics_cpp.InverseKinematics -------------------------------------------
 * srcpy/ik/inverse_kinematics.cpp:20-39 over src/ik/{inverse_kinematics,com_tasks,end_effector_tasks,
 * regularization_costs}.cpp.  x = [q(19), v(18)], u = accelerations (18); frame = index into the
 * model's frame list.  Costs accumulate per node until optimize() and are then discarded. */
typedef struct bmpc_ik bmpc_ik_t;
bmpc_ik_t *bmpc_ik_create(const bmpc_model_t *model, int n_col);                                   /* :21 */
void bmpc_ik_destroy(bmpc_ik_t *h);
int bmpc_ik_n_col(const bmpc_ik_t *h);
int bmpc_ik_setup_costs(bmpc_ik_t *h, const double *dt, int n);                                     /* :22 */
int bmpc_ik_optimize(bmpc_ik_t *h, const double *x0);                                               /* :23 */
int bmpc_ik_get_xs(const bmpc_ik_t *h, double *xs);          /* (n_col+1) x 37                        :24 */
int bmpc_ik_get_us(const bmpc_ik_t *h, double *us);          /* n_col x 18                            :25 */
int bmpc_ik_return_opt_com(bmpc_ik_t *h, double *com);       /* (n_col+1) x 3                         :26 */
int bmpc_ik_return_opt_mom(bmpc_ik_t *h, double *mom);       /* (n_col+1) x 6                         :27 */
int bmpc_ik_add_position_tracking_task(bmpc_ik_t *h, int frame, int sn, int en, const double *traj3,
                                       double wt, const char *name);                               /* :29 */
int bmpc_ik_add_position_tracking_task_single(bmpc_ik_t *h, int frame, const double *traj3, double wt,
                                              const char *name, int time_step);                    /* :30 */
int bmpc_ik_add_terminal_position_tracking_task(bmpc_ik_t *h, int frame, const double *traj3, double wt,
                                                const char *name);                                 /* :31 */
int bmpc_ik_add_velocity_tracking_task(bmpc_ik_t *h);        /* prints "function not implemented"     :32 */
int bmpc_ik_add_com_position_tracking_task(bmpc_ik_t *h, int sn, int en, const double *traj, int rows,
                                           double wt, const char *name, int is_terminal);          /* :33 */
int bmpc_ik_add_centroidal_momentum_tracking_task(bmpc_ik_t *h, int sn, int en, const double *traj, int rows,
                                                  double wt, const char *name, int is_terminal);   /* :34 */
int bmpc_ik_add_state_regularization_cost(bmpc_ik_t *h, int sn, int en, double wt, const char *name,
                                          const double *w36, const double *xreg37, int is_terminal); /* :35 */
int bmpc_ik_add_state_regularization_cost_single(bmpc_ik_t *h, int time_step, double wt, const char *name,
                                                 const double *w36, const double *xreg37);        /* :36 */
int bmpc_ik_add_ctrl_regularization_cost(bmpc_ik_t *h, int sn, int en, double wt, const char *name,
                                         const double *w18, const double *ureg18, int is_terminal); /* :37 */
int bmpc_ik_add_ctrl_regularization_cost_single(bmpc_ik_t *h, int time_step, double wt, const char *name,
                                                const double *w18, const double *ureg18);         /* :38 */
/* additive: DDP telemetry of the last optimize; status 0 converged, 1 maxiter, 2 regularisation maxed */
int bmpc_ik_last_stats(const bmpc_ik_t *h, int *iters, int *status, double *cost, double *stop);

/* batch of independent IK problems (additive).  Device pointers:
 *   x0 [B][37], dt [B][n_col], tasks [B][n_col+1][BMPC_IK_NODE_TASK_DOUBLES] per node:
 *       4 x {weight, frame, ref(3)} | com {weight, ref(3)} | mom {weight, ref(6)} | state weight | ctrl weight
 *   state_w [.][36], ctrl_w [.][18] (batch stride 0 = shared), x_reg [B][37]; with node strides sn_* != 0 these are per
 *       node too: state_w [.][n_col+1][36], x_reg [.][n_col+1][37] (batch stride s_x_reg, 0 = 37), ctrl_w [.][n_col][18]
 *       (what the acyclic generator's time-varying add_*_regularization_cost_single calls produce)
 *   ws [B][bmpc_ik_workspace_doubles(n_col)] scratch + results (offsets: bmpc_ik_layout), active: one int */
#define BMPC_IK_NODE_TASK_DOUBLES 33
/* Scheduling thresholds of ONE batch solve (no effect on results; they replace flipping the process-wide bmpc_ik_set_* defaults
 * around a call, which raced between host threads driving different streams).  Every field: 0 = the process default,
 * < 0 = never, n > 0 = while at most n problems are still iterating.
 * ABI version 2 (bmpc_abi_version): bmpc_ik_batch_t ends in this struct and active_list grew with it.  A caller MUST
 * zero-initialise the whole bmpc_ik_batch_t (memset) before filling it -- a zero sched is "all defaults, no test switch" --,
 * check bmpc_ik_batch_struct_size() == sizeof(bmpc_ik_batch_t), and size active_list with bmpc_ik_active_list_ints(B), never with
 * a formula of its own (the list code writes the express lane's xlist / xmeta behind the two lists). */
typedef struct {
    int spec_below;         /* four step lengths of a problem side by side (bmpc_ik_set_speculative_below) */
    int all_steps_below;    /* all ten step lengths at once on three workgroups (bmpc_ik_set_all_steps) */
    int gains_wave_below;   /* a second wave per problem for the Riccati gains (bmpc_ik_set_gains_wave_below) */
    int express_cap;        /* NOT a threshold: the most problems the express lane may take off the batch (bmpc_ik_set_express_capacity;
                               0 = the process default, < 0 = no express lane) */
    int debug_inject;       /* tests only: 1 = overwrite the first active-list entry with an out-of-range index right after the
                               list is initialised; the solve must then return BMPC_DEVICE_ERROR (index checks of the list code);
                               2 = the express lane takes its problems in front of iteration 2 whatever the batch looks like;
                               any other value is ignored (as 0) */
} bmpc_ik_sched_t;
typedef struct {
    int B, n_col, maxiter;
    const bmpc_model_t *model;
    const double *x0, *dt, *tasks, *state_w, *x_reg, *ctrl_w;
    long s_state_w, s_ctrl_w;
    double *ws;
    int *active;
    int *iters_run;   /* host int or NULL: DDP iterations the loop executed */
    long s_x_reg, sn_state_w, sn_x_reg, sn_ctrl_w;   /* 0 = defaults: x_reg [B][37], one vector per problem */
    int *active_list; /* device, bmpc_ik_active_list_ints(B) ints of scratch, or NULL.  With it every launch of the DDP loop covers
                         only the problems still iterating (an index list the forward pass rebuilds each iteration), and the few
                         problems whose line search goes past four step lengths get all ten at once from then on; without it
                         every launch covers all B and finished problems return at once.  Results do not depend on it. */
    bmpc_ik_sched_t sched;
} bmpc_ik_batch_t;
int bmpc_ik_batch_struct_size(void);     /* sizeof(bmpc_ik_batch_t), to catch binding drift */
/* diagnostics: workgroups of each IK kernel a CU holds at once according to the runtime's occupancy query, in the order calcdiff
 * (two waves per node pair), calcdiff1 (one wave per pair), backward<1>, backward<2>, forward<1>, forward<2>, forward<3>, state */
void bmpc_ik_kernel_occupancy(int *out8);
long bmpc_ik_active_list_ints(long B);   /* length of bmpc_ik_batch_t.active_list */
int bmpc_ik_workspace_doubles(int n_col);
void bmpc_ik_layout(int n_col, long *offsets8);      /* xs, us, scalars, K, k, fs, Lx, Lqq (the q-block of L_xx in MFMA tile layout) */
/* telemetry: rows [iteration i < *iters][*width = 4] at *offset of a problem's workspace: cost, regularisation, accepted step
 * length (0 = none) and stopping criterion |Q_u|^2 as SolverDDP holds them at the end of iteration i */
void bmpc_ik_layout_trace(int n_col, long *offset, int *iters, int *width);
/* Line-search scheduling of the batched DDP (no effect on results): while at most n_active problems are still
 * iterating, four step lengths of a problem are tried side by side (one wave per problem) instead of one after the
 * other (four problems per wave); with at most bmpc_ik_set_all_steps' n_active left (default 0 = never: no measurable gain, EXPERIMENTS.md 9), all ten step
 * lengths at once on three workgroups per problem.  Default 1024 (one wave per SIMD of an MI355X); 0 = never.  Both return
 * the old value. */
int bmpc_ik_set_speculative_below(int n_active);
/* (experiment switch) with MORE than n_active problems iterating -- and at most the bound above -- the side-by-side line search runs
 * every role of a problem on ONE wave instead of two (as many waves as problems: one round over the chip's SIMDs where two waves per
 * problem need two).  Default 0 = never.  Returns the old value. */
int bmpc_ik_set_spec_one_wave_above(int n_active);
int bmpc_ik_set_all_steps(int n_active);
/* Riccati-pass scheduling (no effect on results): while at most n_active problems are still iterating, each gets a second
 * wave that computes and stores the gains K, k one node behind the recursion.  Default 512; 0 = never.  Returns the old value. */
int bmpc_ik_set_gains_wave_below(int n_active);
/* Host waits of the DDP loop (no effect on results): 1 (default) = the waiting host thread sleeps until the device's interrupt
 * (hipEventBlockingSync), 0 = it spins.  Returns the old value. */
int bmpc_ik_set_blocking_waits(int on);
/* The express lane (needs bmpc_ik_batch_t.active_list; no effect on results).  A batch that converges fast leaves a thin tail of
 * problems iterating long after the rest has finished, and in lock-step those pay every early iteration at the whole batch's pace.
 * In front of iterations 2..12 a one-workgroup kernel looks at the batch and, ONCE -- when (almost) nobody has finished yet, at
 * least half of the problems are near the stopping threshold and an iteration earlier (almost) none was -- takes the n problems
 * farthest from it (never more than an eighth of the batch) off the active list; a persistent fused kernel (one four-wave
 * workgroup per problem: derivative pass, Riccati pass and line search of whole DDP iterations without leaving the chip, no
 * host look) runs them to the end on a side stream while the batch goes on without them.  Default 96; 0 = no express lane.
 * Returns the old value. */
int bmpc_ik_set_express_capacity(int n);
/* Batches of at most n problems (the single-problem handles InverseKinematics / KinoDynMP above all) run their whole DDP in ONE
 * launch of the fused kernel, every problem on a CU of its own, with no host look between iterations (no effect on results).
 * Default 16; 0 = never.  Returns the old value. */
int bmpc_ik_set_fused_direct_max(int n);
/* ... and what "near the stopping threshold" means in that trigger: |Q_u|^2 < stop at the problem's last Riccati pass (SolverDDP
 * stops below 1e-9).  Default 1.0.  Returns the old value. */
double bmpc_ik_set_express_near(double stop);
int bmpc_ik_solve_batch_device(const bmpc_ik_batch_t *d, void *hip_stream);
/* Measurement aid (additive): with profiling on, the DDP loop brackets each of its kernels with events; after a batch solve
 * bmpc_ik_last_profile returns the summed milliseconds of ik_state / ik_calcdiff / ik_backward / ik_forward and of the rest
 * of the loop (bench.py's per-kernel split; rocprofv3 --kernel-trace gives the same numbers, profiles/). */
/* self test (host arrays): the state operators diff(x0, x1) and x0 (+) dx of n samples, x [n][37], dx [n][36], by the
 * quaternion versions the forward pass uses (dq [n][36], iq [n][37]) and by the rotation-matrix versions (dr, ir) */
int bmpc_ik_selftest_state_ops(const double *x0, const double *x1, const double *dx, int n, double *dq, double *dr, double *iq, double *ir);
int bmpc_ik_set_profile(int on);            /* returns the old setting */
void bmpc_ik_last_profile(double *ms5);
/* [com, vcom, hg.angular] of x = [q, v]: what KinoDynMP::optimize feeds the centroidal solve (kino_dyn.cpp:42,86-97) */
int bmpc_ik_centroidal_state_device(const bmpc_model_t *model, const double *x, double *out9, int B, void *hip_stream);

/* biconvex_mpc_cpp.KinoDynMP ---------------------------------------------------------
 * srcpy/motion_planner/biconvex.cpp:55-63 over src/motion_planner/kino_dyn.cpp */
typedef struct bmpc_kinodyn bmpc_kinodyn_t;
bmpc_kinodyn_t *bmpc_kinodyn_create(const bmpc_model_t *model, double m, int n_eff, int dyn_col, int ik_col); /* :56 */
void bmpc_kinodyn_destroy(bmpc_kinodyn_t *h);
bmpc_biconvex_t *bmpc_kinodyn_return_dyn(bmpc_kinodyn_t *h);      /* borrowed, owned by h           :57 */
bmpc_ik_t *bmpc_kinodyn_return_ik(bmpc_kinodyn_t *h);             /* borrowed, owned by h           :58 */
int bmpc_kinodyn_optimize(bmpc_kinodyn_t *h, const double *q, const double *v, int dyn_iters, int kino_dyn_iters); /* :59 */
int bmpc_kinodyn_set_com_tracking_weight(bmpc_kinodyn_t *h, double w);                              /* :60 */
int bmpc_kinodyn_set_mom_tracking_weight(bmpc_kinodyn_t *h, double w);                              /* :61 */
int bmpc_kinodyn_compute_solve_times(bmpc_kinodyn_t *h);                                            /* :62 */
int bmpc_kinodyn_return_solve_times(const bmpc_kinodyn_t *h, double *t3);  /* dyn, ik, total seconds  :63 */

/* batch of KinoDynMP::optimize calls (additive), all pointers on the device:
 *   x [B][37] = [q, v];  dyn.x_init [B][9] is OVERWRITTEN with [com, vcom, hg.angular] of x (kino_dyn.cpp:42),
 *   dyn solved (use cold_start = 1 for set_warm_starts' behaviour), then the com / momentum
 *   references of ik.tasks are OVERWRITTEN from dyn.X (kino_dyn.cpp:50-56; their weights are the
 *   caller's wt_com / wt_mom already in the task blocks) and the IK-DDP runs. */
typedef struct {
    bmpc_batch_t dyn;
    bmpc_ik_batch_t ik;
    const double *x;
} bmpc_kinodyn_batch_t;
int bmpc_kinodyn_solve_batch_device(const bmpc_kinodyn_batch_t *d, void *hip_stream);

/* harness inputs on the device (additive; SURVEY 8f-1, centroidal level) -----------------------------
 * What SoloMpcGaitGen.create_cnt_plan / create_costs compute per MPC call (abstract_cyclic_gen.py:159-414, 564-607)
 * for B problems at once, data path only (no MCTS locations, noise or height map): from the current CoM, foot
 * positions, time, desired velocities -> cnt_plan [B][H][4][4], swing_time [B][H][4], dt [B][H], X_nom [B][9H],
 * X_ter [B][9], ready to be handed to bmpc_biconvex_solve_batch_device without leaving HBM.  The gait table is
 * DEVICE memory as well (n_gaits entries; gait_id [B] selects, NULL = entry 0). */
typedef struct {
    double gait_period, gait_dt, gait_horizon, nom_ht;
    double stance_percent[4], phase_offset[4];
    double ori_correction[3];
    double offsets_xy[4][2];     /* hip offsets from the CoM after the harness' rounding and +-0.04 widening (:56-72) */
} bmpc_gait_params_t;
typedef struct {
    int B, n_col, n_gaits, reserved_;
    const bmpc_gait_params_t *gaits;   /* device */
    const int *gait_id;                /* [B] or NULL */
    const double *t0;                  /* [B] */
    const double *com;                 /* [B][3] centre of mass (xy are rounded to 3 decimals inside, :164) */
    const double *feet0;               /* [B][4][3] current foot positions (rounded inside, :215) */
    const double *v_des;               /* [B][3] desired CoM velocity, already in the yaw frame */
    const double *w_des;               /* [B] */
    const double *x_init;              /* [B][9] centroidal state */
    const double *amom;                /* [B][3] orientation-correction momentum (:616-627) or NULL */
    const double *hip_off;             /* [B][4][2] yaw-rotated hip offsets or NULL (gaits[].offsets_xy) */
    double *cnt_plan, *swing_time, *dt, *X_nom, *X_ter;
} bmpc_plan_batch_t;
int bmpc_plan_batch_device(const bmpc_plan_batch_t *d, void *hip_stream);

/* The same for whole-body states: everything SoloMpcGaitGen.optimize prepares before kd.optimize
 * (abstract_cyclic_gen.py:629-663 -> create_cnt_plan, create_costs) from x = [q, v], the time and the desired body-frame
 * velocity, on the device: forward kinematics (CoM, feet, centroidal state), v_des in the world frame, yaw-rotated hip
 * offsets, the orientation-correction momentum log3(R_q^T), then the contact plan / cost references as above and the
 * IK task blocks of bmpc_ik_batch_t (frame tasks from the plan; CoM / momentum weights, their references are filled
 * by bmpc_kinodyn_solve_batch_device).  w_des = 0 (the data path).  Outputs feed bmpc_kinodyn_solve_batch_device. */
typedef struct {
    int B, n_col, ik_col, reserved_;
    const bmpc_model_t *model;
    const bmpc_gait_params_t *gait;    /* device, one entry; offsets_xy in the body frame */
    int foot_frame[4];                 /* frame indices of the end effectors (eff_names) */
    double step_ht, swing_wt[2], cent_wt[2], reg_wt[2];   /* weight_abstract.py fields the task list uses */
    const double *x;                   /* [B][37] */
    const double *t0;                  /* [B] */
    const double *v_des_body;          /* [B][3] */
    /* intermediates (device, caller-allocated): */
    double *com, *feet0, *v_des, *w_des, *hip_off, *amom;   /* [B][3], [B][4][3], [B][3], [B], [B][4][2], [B][3] */
    /* outputs: */
    double *x_init;                    /* [B][9] centroidal state of x */
    double *cnt_plan, *swing_time, *dt, *X_nom, *X_ter;
    double *ik_tasks;                  /* [B][ik_col + 1][33] */
} bmpc_wb_plan_batch_t;
int bmpc_wb_plan_batch_device(const bmpc_wb_plan_batch_t *d, void *hip_stream);

/* The 1 kHz plan of SoloMpcGaitGen.optimize (abstract_cyclic_gen.py:677-692) for a batch, on the device:
 * out[b] = vstack_{i < size} linspace(knots[b][i], knots[b][i+1], int(dt[b][i] / step)), end points included (and so
 * repeated at the seams, as there).  knots [B][n_knots][width], dt [B][dt_stride] (first `size` entries used),
 * out [B][max_rows][width], rows [B] = number of rows written for each problem (<= max_rows; more are dropped). */
typedef struct {
    int B, n_knots, width, size, max_rows, dt_stride;
    double step;
    const double *knots, *dt;
    double *out;
    int *rows;
} bmpc_interp_batch_t;
int bmpc_interp_batch_device(const bmpc_interp_batch_t *d, void *hip_stream);

/* Output stage of the data path: InverseDynamicsController.id_joint_torques
 * (ISL/examples/controllers/robot_id_controller.py:57-86) and the rows the rollout loop records from it
 * (ISL/examples/iterative_algorithm/simulation.py:484-528), for n samples at once, on the device:
 *   tau_ff = (rnea(q_des, v_des, a_des) - sum_j J_j(q_des)^T [f_j; 0])[6:]
 *   tau_fb = -kp (q[7:] - q_des[7:]) - kd (v[6:] - v_des[6:])
 *   action = (tau_ff + tau_fb + kd v[6:]) / kp + q[7:]                     (action_type "pd_target")
 *   state  = [v (18) | q[0:2] - foot_j[0:2], j < 4 (8) | q[2:] (17)]        (43 doubles)
 * Every input is a device array of rows with its own row stride in doubles, so the rows of the 1 kHz plan
 * (xs_int: q at +0, v at +19, stride 37; us_int; f_int) can be passed in place.  q / v NULL = the desired rows
 * (a sample exactly on its plan: tau_fb = 0).  Outputs are dense ([n][12], [n][43]); NULL ones are skipped.
 * Each end effector must hang off a different leg. */
typedef struct {
    long n;
    const bmpc_model_t *model;
    int foot_frame[4];                 /* frame indices of the end effectors, in the order of f */
    double kp[12], kd[12];
    const double *q, *v;               /* measured state: rows of 19 / 18 */
    const double *q_des, *v_des, *a_des, *f;   /* rows of 19 / 18 / 18 / 12 */
    long s_q, s_v, s_q_des, s_v_des, s_a_des, s_f;
    double *tau_ff, *tau_fb, *action, *state;
} bmpc_id_batch_t;
int bmpc_id_batch_device(const bmpc_id_batch_t *d, void *hip_stream);

/* Contact-conditioned perturbation of nominal states: the sampler of the data-collection loop
 * (ISL/examples/iterative_algorithm/data_collection.py:188-262) for B nominal states at once, on the device.
 * For state b the draws z[b][0..K) (standard normal, 36 each: 18 for the position, 18 for the velocity perturbation)
 * are consumed in order until one gives a configuration with no foot below the ground, as the reference's while loop
 * does: pos = mu + sigma z (mu / sigma = base position, base orientation, joint position, velocity),
 * q' = integrate(q, (I - pinv(J) J) pos), v' = v + (I - pinv(J * vel) (J * vel)) pos with J the stacked linear
 * Jacobians of the feet whose contact flag is 1.  chosen[b] = index of the accepted draw, or -1 when all K were
 * rejected (q_out / v_out then hold the nominal state and the caller draws again for that state).
 * contact: flag of foot e of state b at contact[b * s_contact_b + e * s_contact_e], so a contact-plan row
 * (cnt_plan[b][knot][e][0]) can be passed in place. */
typedef struct {
    int B, K;
    const bmpc_model_t *model;
    int foot_frame[4];
    double mu[4], sigma[4];
    const double *q, *v;               /* [B][19], [B][18] nominal states */
    const double *contact;
    long s_contact_b, s_contact_e;
    const double *z;                   /* [B][K][36] */
    double *q_out, *v_out;             /* [B][19], [B][18] */
    int *chosen;                       /* [B] */
} bmpc_perturb_batch_t;
int bmpc_perturb_batch_device(const bmpc_perturb_batch_t *d, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif

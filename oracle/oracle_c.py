"""ctypes access to oracle/liboracle_biconvex.so (the C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by bunmpc_amd.  PARITY UNPINNED (see biconvex_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_biconvex.so")
NSTATS = 6


class Params(C.Structure):
    _fields_ = [("rho", C.c_double), ("beta", C.c_double), ("mu", C.c_double),
                ("tol", C.c_double), ("exit_tol", C.c_double), ("maxit", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "biconvex_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_biconvex.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_gait_phi.restype = C.c_double
        _lib.orc_gait_phi.argtypes = [C.c_double] * 3
        _lib.orc_gait_phase.restype = C.c_int
        _lib.orc_gait_phase.argtypes = [C.c_double] * 4
        _lib.orc_gait_percent_in_phase.restype = C.c_double
        _lib.orc_gait_percent_in_phase.argtypes = [C.c_double] * 4
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def default_params(rho=5e4):
    p = Params()
    lib().orc_default_params(C.byref(p))
    p.rho = rho
    return p


def create_bound_constraints(cnt_plan, b):
    H, E, _ = cnt_plan.shape
    cnt_plan, b = _f64(cnt_plan), _f64(b)
    lb = np.empty(9 * (H + 1)); ub = np.empty(9 * (H + 1))
    lib().orc_create_bound_constraints(H, E, _p(cnt_plan), _p(b), _p(lb), _p(ub))
    return lb, ub


def create_cost_X(W_X, W_X_ter, X_ter, X_nom):
    H = W_X.shape[0] // 9
    Qx = np.empty(9 * (H + 1)); qx = np.empty(9 * (H + 1))
    lib().orc_create_cost_X(H, _p(_f64(W_X)), _p(_f64(W_X_ter)), _p(_f64(X_ter)), _p(_f64(X_nom)),
                            _p(Qx), _p(qx))
    return Qx, qx


def dense_A_x(cnt_plan, dt, m, X):
    H, E, _ = cnt_plan.shape
    A = np.empty((9 * (H + 1), 3 * E * H)); b = np.empty(9 * (H + 1))
    lib().orc_dense_A_x(H, E, C.c_double(m), _p(_f64(cnt_plan)), _p(_f64(dt)), _p(_f64(X)), _p(A), _p(b))
    return A, b


def dense_A_f(cnt_plan, dt, m, F, x_init):
    H, E, _ = cnt_plan.shape
    n = 9 * (H + 1)
    A = np.empty((n, n)); b = np.empty(n)
    lib().orc_dense_A_f(H, E, C.c_double(m), _p(_f64(cnt_plan)), _p(_f64(dt)), _p(_f64(F)),
                        _p(_f64(x_init)), _p(A), _p(b))
    return A, b


def biconvex_solve(cnt_plan, dt, m, x_init, Qx, qx, Qf, lbx, ubx, X, F, P, L_x=2.25e6,
                   L_f=506.25, rho=5e4, num_iters=10, maxit=150, tol=1e-5, exit_tol=1e-3,
                   beta=1.5, mu=1.0, qf=None):
    """One BiConvexMP::optimize; same signature/return as oracle_np.biconvex_solve."""
    cnt_plan = _f64(cnt_plan)
    H, E, _ = cnt_plan.shape
    prm = Params(rho, beta, mu, tol, exit_tol, maxit)
    X, F, P = _f64(X).copy(), _f64(F).copy(), _f64(P).copy()
    Lx, Lf = C.c_double(L_x), C.c_double(L_f)
    hist = np.full(num_iters, np.nan)
    stats = np.zeros(NSTATS, dtype=np.int32)
    args = [_f64(a) for a in (dt, x_init, Qx, qx, Qf, qf, lbx, ubx)]
    lib().orc_biconvex_solve(H, E, C.c_double(m), C.byref(prm), _p(cnt_plan), _p(args[0]),
                             _p(args[1]), _p(args[2]), _p(args[3]), _p(args[4]), _p(args[5]),
                             _p(args[6]), _p(args[7]), _p(X), _p(F), _p(P), C.byref(Lx),
                             C.byref(Lf), num_iters, _p(hist), _p(stats))
    return dict(X=X, F=F, P=P, L_x=Lx.value, L_f=Lf.value, hist=hist[:stats[0]],
                stats=stats.astype(np.int64))


def solve_batch(batch, num_iters=10, maxit=150, tol=1e-5, exit_tol=1e-3, nthreads=0,
                L_x=2.25e6, L_f=506.25, warm=None, fast=False, trace=False, x_init=None):
    """Solve a bunmpc_amd.problems.Batch with the C oracle (cold start unless warm=(X,F,P)).
    fast=True: the matrix-free variant (biconvex_fast.c) instead of the strict restatement.
    trace=True: also `hist` [B][num_iters] (||A_f X - b_f|| after every ADMM iteration, NaN where none ran) and `trace`
    [B][num_iters][4] (running totals F-iterations, X-iterations, F-retries, X-retries; -1 where none ran).
    x_init: replaces batch.x_init (the perturbed members of tools/chaos_ensemble.py); the cold start tiles it like the batch's."""
    B, H, E = batch.B, batch.H, batch.E
    nx, nf = 9 * (H + 1), 3 * E * H
    Qx = np.empty((B, nx)); qx = np.empty((B, nx)); lbx = np.empty((B, nx)); ubx = np.empty((B, nx))
    Qf = np.empty((B, nf))
    for b in range(B):
        sb = 0 if batch.W_X.shape[0] == 1 else b
        Qx[b], qx[b] = create_cost_X(batch.W_X[sb], batch.W_X_ter[sb], batch.X_ter[b], batch.X_nom[b])
        lbx[b], ubx[b] = create_bound_constraints(batch.cnt_plan[b], batch.bounds[0 if batch.bounds.shape[0] == 1 else b])
        Qf[b] = batch.W_F[0 if batch.W_F.shape[0] == 1 else b]
    X, F, P = batch.warm_start() if warm is None else [np.array(a, dtype=np.float64) for a in warm]
    X, F, P = _f64(X), _f64(F), _f64(P)
    if x_init is not None and warm is None:      # KinoDynMP::set_warm_starts tiles x_init (kino_dyn.cpp:83-99)
        X = np.ascontiguousarray(np.tile(np.asarray(x_init, np.float64), (1, H + 1)))
    Lx = np.full(B, L_x); Lf = np.full(B, L_f)
    stats = np.zeros((B, NSTATS), dtype=np.int32)
    prm = Params(batch.rho, 1.5, getattr(batch, 'mu', 1.0), tol, exit_tol, maxit)
    cnt, dt, xi = _f64(batch.cnt_plan), _f64(batch.dt), _f64(batch.x_init if x_init is None else x_init)
    hist = np.full((B, max(num_iters, 1)), np.nan) if trace else None
    tr = np.full((B, max(num_iters, 1), 4), -1, dtype=np.int32) if trace else None
    fn = lib().orc_fast_solve_batch_traced if fast else lib().orc_biconvex_solve_batch_traced
    ndiv = fn(B, H, E, C.c_double(batch.m), C.byref(prm), _p(cnt), _p(dt),
              _p(xi), _p(Qx), _p(qx), _p(Qf), None, _p(lbx), _p(ubx), 0,
              _p(X), _p(F), _p(P), _p(Lx), _p(Lf), num_iters, _p(stats), nthreads, _p(hist), _p(tr))
    out = dict(X=X, F=F, P=P, L_x=Lx, L_f=Lf, stats=stats.astype(np.int64), n_diverged=ndiv,
               Qx=Qx, qx=qx, Qf=Qf, lbx=lbx, ubx=ubx)
    if trace:
        out["hist"], out["trace"] = hist, tr.astype(np.int64)
    return out

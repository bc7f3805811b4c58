"""Batched BiConvex MPC solves on one MI355X: B independent `BiconvexMP.optimize` calls in
one kernel launch (bmpc_biconvex_solve_batch_device / _host in include/bunmpc.h).

`DeviceBatch` keeps a `problems.Batch` resident in HBM as torch tensors (torch is used for
device memory and streams only) and launches on torch's current stream, so
`torch.cuda.Event`s bracket the kernel correctly.  `solve_host` is the numpy-in /
numpy-out path that needs no torch.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import L0_F, L0_X, NSTATS


def _solver_fields(desc, batch, num_iters, maxit, tol, exit_tol, beta, mu):
    mu = batch.mu if mu is None else mu
    desc.B, desc.n_col, desc.n_eff = batch.B, batch.H, batch.E
    desc.num_iters, desc.maxit = num_iters, maxit
    desc.m, desc.rho, desc.mu, desc.beta, desc.tol, desc.exit_tol = batch.m, batch.rho, mu, beta, tol, exit_tol


def _stride(a):
    """batch stride in doubles of a (1 or B, ...) array: 0 when shared"""
    return 0 if a.shape[0] == 1 else int(np.prod(a.shape[1:]))


def algorithmic_bytes_per_solve(H, E=4, per_problem_weights=False):
    """inputs (4E+10)H+18 doubles, outputs 18(H+1)+3EH+1 doubles
    (+ 9+9+3E+6+3 doubles when weights are per problem)."""
    n = (4 * E + 10) * H + 18 + 18 * (H + 1) + 3 * E * H + 1
    if per_problem_weights:
        n += 9 + 9 + 3 * E + 6 + 3
    return 8 * n


class DeviceBatch:
    """A problems.Batch resident on one GPU, harness form (the kernel applies create_cost_X /
    create_cost_F / create_bound_constraints itself)."""

    def __init__(self, batch, device="cuda", num_iters=10, maxit=150, tol=1e-5, exit_tol=1e-3,
                 beta=1.5, mu=None, keep_hist=False, precision="f64", plan=None):
        """plan: a plan_batch.DevicePlan whose tensors (cnt_plan, dt, X_nom, X_ter, x_init) are used in place of the
        batch's host arrays -- inputs built on the GPU never leave HBM"""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceBatch needs a GPU: no CPU fallback exists for the solve")
        self.torch = torch
        self.batch = batch
        self.device = torch.device(device)
        self.num_iters = num_iters
        B, H, E = batch.B, batch.H, batch.E
        f64 = torch.float64

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

        if plan is None:
            self.t = dict(cnt_plan=up(batch.cnt_plan), dt=up(batch.dt), x_init=up(batch.x_init),
                          X_nom=up(batch.X_nom), X_ter=up(batch.X_ter))
        else:
            assert plan.B == B and plan.H == H
            self.t = dict(cnt_plan=plan.cnt_plan, dt=plan.dt, x_init=plan.inp["x_init"], X_nom=plan.X_nom, X_ter=plan.X_ter)
        self.t.update(W_X=up(batch.W_X), W_X_ter=up(batch.W_X_ter), W_F=up(batch.W_F), bounds=up(batch.bounds))
        self.X = torch.empty((B, 9 * (H + 1)), dtype=f64, device=self.device)
        self.F = torch.empty((B, 3 * E * H), dtype=f64, device=self.device)
        self.P = torch.empty((B, 9 * (H + 1)), dtype=f64, device=self.device)
        self.L_x = torch.full((B,), L0_X, dtype=f64, device=self.device)
        self.L_f = torch.full((B,), L0_F, dtype=f64, device=self.device)
        self.dyn_viol = torch.zeros(B, dtype=f64, device=self.device)
        self.stats = torch.zeros((B, NSTATS), dtype=torch.int32, device=self.device)
        self.hist = torch.full((B, max(num_iters, 1)), float("nan"), dtype=f64,
                               device=self.device) if keep_hist else None
        # with keep_hist the solve also leaves its discrete path per ADMM iteration (bmpc_batch_t.trace; -1 where none ran)
        self.trace = torch.full((B, max(num_iters, 1), 4), -1, dtype=torch.int32, device=self.device) if keep_hist else None
        d = _lib.Batch()
        _lib.lib().bmpc_batch_defaults(C.byref(d))
        _solver_fields(d, batch, num_iters, maxit, tol, exit_tol, beta, mu)
        d.precision = {"f64": 0, "f32": 1}[precision]
        d.raw = 0
        d.cold_start = 1
        for k, v in self.t.items():
            setattr(d, k, v.data_ptr())
        d.sW_X, d.sW_X_ter = _stride(batch.W_X), _stride(batch.W_X_ter)
        d.sW_F, d.sbounds = _stride(batch.W_F), _stride(batch.bounds)
        d.X, d.F, d.P = self.X.data_ptr(), self.F.data_ptr(), self.P.data_ptr()
        d.L_x, d.L_f = self.L_x.data_ptr(), self.L_f.data_ptr()
        d.dyn_viol, d.stats = self.dyn_viol.data_ptr(), self.stats.data_ptr()
        d.hist = self.hist.data_ptr() if keep_hist else None
        d.trace = self.trace.data_ptr() if keep_hist else None
        self.desc = d

    def set_warm_start(self, X, F, P, L_x=None, L_f=None):
        """set_warm_start_vars for the whole batch; the next solve() starts from these."""
        torch = self.torch
        self.X.copy_(torch.as_tensor(np.asarray(X), dtype=torch.float64))
        self.F.copy_(torch.as_tensor(np.asarray(F), dtype=torch.float64))
        self.P.copy_(torch.as_tensor(np.asarray(P), dtype=torch.float64))
        self.L_x.fill_(L0_X) if L_x is None else self.L_x.copy_(torch.as_tensor(np.asarray(L_x)))
        self.L_f.fill_(L0_F) if L_f is None else self.L_f.copy_(torch.as_tensor(np.asarray(L_f)))
        self.desc.cold_start = 0

    def cold_start(self, carry_step_constants=False):
        """Every solve starts as KinoDynMP::set_warm_starts does (kino_dyn.cpp:83-99): X = tile(x_init), F = 0, P = 0.
        carry_step_constants=False: with the constructor's FISTA constants too (a fresh KinoDynMP per problem: independent batch
        elements).  True: L_x / L_f stay what the previous solve left (or set_step_constants set) -- the reference never resets
        FISTA's L_ between the optimize calls of one object (fista.hpp:52), so this is the mode for successive replans of the same
        rollouts."""
        self.desc.cold_start = 2 if carry_step_constants else 1

    def set_step_constants(self, L_x, L_f):
        torch = self.torch
        self.L_x.copy_(torch.as_tensor(np.broadcast_to(np.asarray(L_x, dtype=np.float64), self.L_x.shape).copy()))
        self.L_f.copy_(torch.as_tensor(np.broadcast_to(np.asarray(L_f, dtype=np.float64), self.L_f.shape).copy()))

    def solve(self):
        """Asynchronous: one launch on torch's current stream."""
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        if self.hist is not None:      # rows of ADMM iterations that do not run keep their NaN / -1
            self.hist.fill_(float("nan"))
            self.trace.fill_(-1)
        _lib.check(_lib.lib().bmpc_biconvex_solve_batch_device(C.byref(self.desc), C.c_void_p(stream)))

    def results(self):
        self.torch.cuda.synchronize(self.device)
        out = dict(X=self.X.cpu().numpy(), F=self.F.cpu().numpy(), P=self.P.cpu().numpy(),
                   L_x=self.L_x.cpu().numpy(), L_f=self.L_f.cpu().numpy(),
                   dyn_viol=self.dyn_viol.cpu().numpy(), stats=self.stats.cpu().numpy().astype(np.int64))
        if self.hist is not None:
            out["hist"] = self.hist.cpu().numpy()
            out["trace"] = self.trace.cpu().numpy().astype(np.int64)
        return out


def solve_host(batch, num_iters=10, maxit=150, tol=1e-5, exit_tol=1e-3, beta=1.5, mu=None,
               warm=None, L_x=None, L_f=None, raw=None, keep_hist=False, precision="f64"):
    """numpy in / numpy out through bmpc_biconvex_solve_batch_host (copies in, one launch,
    copies out).  warm = (X, F, P) or None for a cold start.  raw = dict(Qx,qx,lbx,ubx,Qf[,qf])
    switches to the raw cost/bound form."""
    B, H, E = batch.B, batch.H, batch.E
    nx, nf = 9 * (H + 1), 3 * E * H
    keep = []

    def f64(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a

    d = _lib.Batch()
    _lib.lib().bmpc_batch_defaults(C.byref(d))
    _solver_fields(d, batch, num_iters, maxit, tol, exit_tol, beta, mu)
    d.precision = {"f64": 0, "f32": 1}[precision]
    for k in ("cnt_plan", "dt", "x_init"):
        setattr(d, k, f64(getattr(batch, k)).ctypes.data)
    if raw is None:
        d.raw = 0
        for k in ("W_X", "W_X_ter", "W_F", "bounds", "X_nom", "X_ter"):
            setattr(d, k, f64(getattr(batch, k)).ctypes.data)
        d.sW_X, d.sW_X_ter = _stride(batch.W_X), _stride(batch.W_X_ter)
        d.sW_F, d.sbounds = _stride(batch.W_F), _stride(batch.bounds)
    else:
        d.raw = 1
        for k in ("Qx", "qx", "lbx", "ubx", "Qf"):
            a = f64(raw[k])
            assert a.shape == (B, nf if k == "Qf" else nx), k
            setattr(d, k, a.ctypes.data)
        if raw.get("qf") is not None:
            d.qf = f64(raw["qf"]).ctypes.data
    if warm is None:
        d.cold_start = 1
        X, F, P = np.zeros((B, nx)), np.zeros((B, nf)), np.zeros((B, nx))
    else:
        d.cold_start = 0
        X, F, P = (np.array(a, dtype=np.float64, order="C").reshape(B, -1) for a in warm)
    Lx = np.full(B, L0_X) if L_x is None else np.array(L_x, dtype=np.float64).reshape(B)
    Lf = np.full(B, L0_F) if L_f is None else np.array(L_f, dtype=np.float64).reshape(B)
    viol = np.zeros(B)
    stats = np.zeros((B, NSTATS), dtype=np.int32)
    hist = np.full((B, max(num_iters, 1)), np.nan) if keep_hist else None
    d.X, d.F, d.P = X.ctypes.data, F.ctypes.data, P.ctypes.data
    d.L_x, d.L_f, d.dyn_viol, d.stats = Lx.ctypes.data, Lf.ctypes.data, viol.ctypes.data, stats.ctypes.data
    trace = np.full((B, max(num_iters, 1), 4), -1, dtype=np.int32) if keep_hist else None
    d.hist = hist.ctypes.data if keep_hist else None
    d.trace = trace.ctypes.data if keep_hist else None
    _lib.check(_lib.lib().bmpc_biconvex_solve_batch_host(C.byref(d)))
    out = dict(X=X, F=F, P=P, L_x=Lx, L_f=Lf, dyn_viol=viol, stats=stats.astype(np.int64))
    if keep_hist:
        out["hist"] = hist
        out["trace"] = trace.astype(np.int64)
    return out

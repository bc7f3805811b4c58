"""oracle/oracle_np.py -- second, independent CPU restatement (numpy + scipy.sparse)
of the reference's centroidal bi-convex ADMM solve.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden generators, never by
the product package).  PARITY UNPINNED: the reference ships no golden vectors for
this path and cannot be built/imported here; this file exists so the
C restatement (oracle/biconvex_oracle.c) is cross-checked by a differently written
implementation of the same reference formulas.

Reference (relative to /root/reference/iterative_supervised_learning/):
  src/dynamics/centroidal.cpp:57-84 (A_x,b_x), :6-37,86-127 (A_f,b_f),
  include/dynamics/centroidal.hpp:22-27 (x_init rows),
  src/solvers/problem.cpp:31-56, src/solvers/fista.cpp:6-70,
  src/motion_planner/biconvex.cpp:27-120.
"""
import numpy as np
import scipy.sparse as sp

G = 9.81


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def build_A_x(X, cnt_plan, dt, m):
    """centroidal.cpp:57-84.  cnt_plan: (H,E,4); returns (A_x sparse, b_x)."""
    H, E, _ = cnt_plan.shape
    A = sp.lil_matrix((9 * (H + 1), 3 * E * H))
    b = np.zeros(9 * (H + 1))
    Xk = X.reshape(H + 1, 9)
    for t in range(H):
        b[9 * t + 3:9 * t + 9] = Xk[t + 1, 3:9] - Xk[t, 3:9]
        b[9 * t + 5] += G * dt[t]
        for n in range(E):
            c = cnt_plan[t, n, 0]
            p = Xk[t, 0:3] - cnt_plan[t, n, 1:4]
            cols = slice(3 * E * t + 3 * n, 3 * E * t + 3 * n + 3)
            A[9 * t + 3:9 * t + 6, cols] = c * (dt[t] / m) * np.eye(3)
            # rows 6..8: dt * c * (f x p) = -dt c [p]x f
            A[9 * t + 6:9 * t + 9, cols] = -c * dt[t] * skew(p)
    return A.tocsc(), b


def build_A_f(F, cnt_plan, dt, m, x_init):
    """centroidal.cpp:6-37,86-127 + centroidal.hpp:22-27."""
    H, E, _ = cnt_plan.shape
    n = 9 * (H + 1)
    A = sp.lil_matrix((n, n))
    b = np.zeros(n)
    Fk = F.reshape(H, E, 3)
    for t in range(H):
        A[9 * t:9 * t + 9, 9 * t:9 * t + 9] = np.eye(9)
        blk = -np.eye(9)
        blk[0:3, 3:6] = dt[t] * np.eye(3)
        A[9 * t:9 * t + 9, 9 * (t + 1):9 * (t + 1) + 9] = blk
        c = cnt_plan[t, :, 0]
        S = (c[:, None] * Fk[t]).sum(axis=0)
        # L_t - L_{t+1} + dt (S x com_t) = dt sum c (f x r)  [rows 6..8, cols 0..2]
        Sx = skew(S) * dt[t]
        for i in range(3):
            for j in range(3):
                if i != j:
                    A[9 * t + 6 + i, 9 * t + j] = Sx[i, j]
        b[9 * t + 3:9 * t + 6] = -S * dt[t] / m
        b[9 * t + 5] += G * dt[t]
        tau = np.zeros(3)
        for k in range(E):
            tau += c[k] * np.cross(Fk[t, k], cnt_plan[t, k, 1:4])
        b[9 * t + 6:9 * t + 9] = tau * dt[t]
    A[9 * H:9 * H + 9, 0:9] = np.eye(9)
    b[9 * H:] = x_init
    return A.tocsc(), b


class Problem:
    """function::ProblemData with a diagonal Q (problem.cpp)."""

    def __init__(self, Qd, q, lb=None, ub=None):
        self.Qd, self.q, self.lb, self.ub = Qd, q, lb, ub
        self.x = None

    def set_data(self, A, b, P, rho):
        self.A, self.rho = A, rho
        self.ATA = 2.0 * (sp.diags(self.Qd) + rho * (A.T @ A))
        self.bPk = -b + P
        self.ATbPk = 2.0 * rho * (A.T @ self.bPk) + self.q

    def grad(self, y):
        return self.ATA @ y + self.ATbPk

    def obj_diff(self, y1, y0):
        d = y1 - y0
        return ((y1 + y0) * self.Qd) @ d + self.q @ d + self.rho * (
            np.sum((self.A @ y1 + self.bPk) ** 2) - np.sum((self.A @ y0 + self.bPk) ** 2))


def soc_projection(v, mu):
    """fista.cpp:52-70, vectorised over the 3-vectors."""
    y = v.reshape(-1, 3).copy()
    s = y[:, 0] ** 2 + y[:, 1] ** 2
    z = y[:, 2].copy()
    zero = (s * mu < -z) | (z < 0)
    cone = (~zero) & (s > mu * z)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = ((mu * mu) * s + (mu * z)) / (((mu * mu) + 1) * s)
    y[cone, 0] *= k[cone]
    y[cone, 1] *= k[cone]
    y[cone, 2] = (mu * s[cone] + z[cone]) / ((mu * mu) + 1)
    y[zero] = 0.0
    return y.reshape(-1)


class Fista:
    def __init__(self, L0, beta=1.5, mu=1.0, soc=False):
        self.L, self.beta, self.mu, self.soc = L0, beta, mu, soc
        self.n_bt = 0

    def step(self, p, y):
        g = p.grad(y)
        while True:
            y1 = y - g / self.L
            if self.soc:
                y1 = soc_projection(y1, self.mu)
            else:
                y1 = np.maximum(np.minimum(y1, p.ub), p.lb)
            d = y1 - y
            G = np.sqrt(d @ d)
            if p.obj_diff(y1, y) > g @ d + (self.L / 2) * (G * G):
                self.L *= self.beta
                self.n_bt += 1
            else:
                return y1, G

    def optimize(self, p, maxit, tol):
        y = p.x.copy()
        t = 1.0
        its = 0
        for _ in range(maxit):
            x1, G = self.step(p, y)
            its += 1
            t1 = 1.0 + np.sqrt(1 + 4 * t * t) / 2.0  # sic (fista.cpp:34)
            y1 = x1 + ((t - 1) / t1) * (x1 - p.x)
            p.x = x1
            if G < tol:
                break
            y, t = y1, t1
        return its


def biconvex_solve(cnt_plan, dt, m, x_init, Qx, qx, Qf, lbx, ubx, X, F, P,
                   L_x=2.25e6, L_f=506.25, rho=5e4, num_iters=10, maxit=150, tol=1e-5,
                   exit_tol=1e-3, beta=1.5, mu=1.0, qf=None):
    """BiConvexMP::optimize (biconvex.cpp:80-120).  Returns dict with X,F,P,L_x,L_f,hist,stats."""
    cnt_plan = np.asarray(cnt_plan, float)
    H, E, _ = cnt_plan.shape
    if qf is None:
        qf = np.zeros(3 * E * H)
    px = Problem(np.asarray(Qx, float), np.asarray(qx, float), np.asarray(lbx, float),
                 np.asarray(ubx, float))
    pf = Problem(np.asarray(Qf, float), np.asarray(qf, float))
    px.x = np.array(X, float)
    pf.x = np.array(F, float)
    P = np.array(P, float)
    fx = Fista(L_x, beta, mu, soc=False)
    ff = Fista(L_f, beta, mu, soc=True)
    hist, trace, it_f, it_x, status, n_admm = [], [], 0, 0, 0, 0
    for _ in range(num_iters):
        A, b = build_A_x(px.x, cnt_plan, dt, m)
        pf.set_data(A, b, P, rho)
        it_f += ff.optimize(pf, maxit, tol)
        A, b = build_A_f(pf.x, cnt_plan, dt, m, x_init)
        px.set_data(A, b, P, rho)
        it_x += fx.optimize(px, maxit, tol)
        viol = A @ px.x - b
        P = P + viol
        nrm = float(np.sqrt(viol @ viol))
        hist.append(nrm)
        trace.append((it_f, it_x, ff.n_bt, fx.n_bt))
        n_admm += 1
        if np.isnan(nrm):
            status = 2
            break
        if nrm < exit_tol:
            break
    return dict(X=px.x, F=pf.x, P=P, L_x=fx.L, L_f=ff.L, hist=np.array(hist), trace=np.array(trace, dtype=np.int64).reshape(-1, 4),
                stats=np.array([n_admm, it_f, it_x, ff.n_bt, fx.n_bt, status]))


def create_bound_constraints(cnt_plan, b):
    """biconvex.cpp:27-55 (X part)."""
    H, E, _ = cnt_plan.shape
    lb = np.full(9 * (H + 1), -np.inf)
    ub = np.full(9 * (H + 1), np.inf)
    for i in range(H):
        if cnt_plan[i, :, 0].sum() > 0:
            lb[9 * i:9 * i + 3] = cnt_plan[i, :, 1:4].max(axis=0) + b[i, 0:3]
            ub[9 * i:9 * i + 3] = cnt_plan[i, :, 1:4].min(axis=0) + b[i, 3:6]
    return lb, ub


def create_cost_X(W_X, W_X_ter, X_ter, X_nom):
    """biconvex.cpp:57-72."""
    Qx = np.concatenate([W_X, W_X_ter])
    qx = np.concatenate([-2 * X_nom * W_X, -2 * X_ter * W_X_ter])
    return Qx, qx

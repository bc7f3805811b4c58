"""oracle/ik_ddp_np.py -- numpy restatement of the reference's whole-body IK:
ik::InverseKinematics (ISL/src/ik/inverse_kinematics.cpp:6-71), its cost adders
(ISL/src/ik/{com_tasks,end_effector_tasks,regularization_costs}.cpp), the "kinematic" action model
(ISL/src/ik/action_model.cpp:43-94: xout = u, Fx = 0, Fu = I) and what it delegates to
crocoddyl 1.9.0 (IntegratedActionModelEuler, CostModelResidual / ActivationModel(Weighted)Quad,
ResidualModel{FrameTranslation,CoMPosition,CentroidalMomentum,State,Control}, SolverDDP::solve with
all defaults).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: crocoddyl 1.9.0 / pinocchio 2.6.9
(.devcontainer/Dockerfile:84-92) are not in /root/reference and not installable here; the solver
semantics below restate the published crocoddyl 1.9.0 algorithm (solver-ddp.cpp: regularisation
x/÷10 in [1e-9, 1e9], step lengths 2^-k k=0..9, acceptance dV > 0.1 dVexp, stop < 1e-9, cold start
at the state's zero with gaps) and are anchored only on the reference's call sites.
"""
import numpy as np

from . import rbd_np as rb


class IKProblem:
    """What the reference accumulates in rcost_arr_[t] / tcost_model_ between two optimize calls."""

    def __init__(self, model, n_col):
        self.model, self.T = model, n_col
        self.dt = np.zeros(n_col)
        self.costs = [dict() for _ in range(n_col + 1)]     # name -> (kind, weight, payload); index T = terminal

    def _add(self, t, name, item):
        if name in self.costs[t]:
            print("Warning: we couldn't add the %s cost item, it already existed." % name)   # crocoddyl addCost
            return
        self.costs[t][name] = item

    # end_effector_tasks.cpp:8-37
    def add_position_tracking_task(self, frame, sn, en, traj, wt, name):
        for i in range(sn, en):
            self._add(i, name + str(i), ("frame", wt, (frame, np.asarray(traj, float).reshape(3))))

    def add_position_tracking_task_single(self, frame, traj, wt, name, time_step):
        self._add(time_step, name, ("frame", wt, (frame, np.asarray(traj, float).reshape(3))))

    def add_terminal_position_tracking_task(self, frame, traj, wt, name):
        self._add(self.T, name, ("frame", wt, (frame, np.asarray(traj, float).reshape(3))))

    # com_tasks.cpp:8-50
    def add_com_position_tracking_task(self, sn, en, traj, wt, name, is_terminal=False):
        traj = np.atleast_2d(np.asarray(traj, float))
        if not is_terminal:
            for i in range(sn, en):
                self._add(i, name, ("com", wt, traj[i - sn]))
        else:
            self._add(self.T, name, ("com", wt, traj[0]))

    def add_centroidal_momentum_tracking_task(self, sn, en, traj, wt, name, is_terminal=False):
        traj = np.atleast_2d(np.asarray(traj, float))
        if not is_terminal:
            for i in range(sn, en):
                self._add(i, name, ("mom", wt, traj[i - sn]))
        else:
            self._add(self.T, name, ("mom", wt, traj[0]))

    # regularization_costs.cpp:8-93
    def add_state_regularization_cost(self, sn, en, wt, name, state_weights, x_reg, is_terminal=False):
        item = ("state", wt, (np.asarray(state_weights, float), np.asarray(x_reg, float)))
        if not is_terminal:
            for i in range(sn, en):
                self._add(i, name, item)
        else:
            self._add(self.T, name, item)

    def add_state_regularization_cost_single(self, time_step, wt, name, state_weights, x_reg):
        self._add(time_step, name, ("state", wt, (np.asarray(state_weights, float), np.asarray(x_reg, float))))

    def add_ctrl_regularization_cost(self, sn, en, wt, name, ctrl_weights, u_reg, is_terminal=False):
        item = ("ctrl", wt, np.asarray(ctrl_weights, float))        # u_reg is ignored (ref = 0), as in the reference
        if not is_terminal:
            for i in range(sn, en):
                self._add(i, name, item)
        else:
            self._add(self.T, name, item)

    def add_ctrl_regularization_cost_single(self, time_step, wt, name, ctrl_weights, u_reg):
        self._add(time_step, name, ("ctrl", wt, np.asarray(ctrl_weights, float)))

    def setup_costs(self, dt):
        self.dt = np.asarray(dt, float).copy()


# --------------------------------------------------------------------- node model ---
def node_calc(prob, t, x, u, diff=False):
    """cost (and derivatives) of node t at (x, u), and the next state for running nodes.
    running nodes: IntegratedActionModelEuler(dt) around xout = u; terminal: unscaled, u = 0."""
    model = prob.model
    nq, nv = model.nq, model.nv
    ndx = 2 * nv
    q, v = x[:nq], x[nq:]
    terminal = t == prob.T
    kin = rb.Kin(model, q, v)
    cost = 0.0
    Lx, Lxx = np.zeros(ndx), np.zeros((ndx, ndx))
    Lu, Luu = np.zeros(nv), np.zeros((nv, nv))
    for kind, wt, pl in prob.costs[t].values():
        if kind == "frame":
            r = kin.frame_placement(pl[0])[1] - pl[1]
            cost += wt * 0.5 * (r @ r)
            if diff:
                Rx = np.zeros((3, ndx)); Rx[:, :nv] = kin.frame_jacobian_lin(pl[0])
                Lx += wt * (Rx.T @ r); Lxx += wt * (Rx.T @ Rx)
        elif kind == "com":
            r = kin.com - pl
            cost += wt * 0.5 * (r @ r)
            if diff:
                Rx = np.zeros((3, ndx)); Rx[:, :nv] = kin.jacobian_com()
                Lx += wt * (Rx.T @ r); Lxx += wt * (Rx.T @ Rx)
        elif kind == "mom":
            r = kin.centroidal_momentum() - pl
            cost += wt * 0.5 * (r @ r)
            if diff:
                Rx = np.hstack([kin.dh_dq(), kin.centroidal_map()])
                Lx += wt * (Rx.T @ r); Lxx += wt * (Rx.T @ Rx)
        elif kind == "state":
            w, xref = pl
            r = rb.state_diff(model, xref, x)
            cost += wt * 0.5 * (w @ (r * r))
            if diff:
                Rx = rb.state_jdiff_second(model, xref, x)
                Lx += wt * (Rx.T @ (w * r)); Lxx += wt * (Rx.T @ (w[:, None] * Rx))
        elif kind == "ctrl":
            uu = np.zeros(nv) if terminal else u
            cost += wt * 0.5 * (pl @ (uu * uu))
            if diff:
                Lu += wt * (pl * uu); Luu += wt * np.diag(pl)
    out = dict(cost=cost)
    if not terminal:
        dt = prob.dt[t]
        dx = np.concatenate([v * dt + u * dt * dt, u * dt])
        out["xnext"] = rb.state_integrate(model, x, dx)
        out["cost"] = dt * cost
        if diff:
            J1, J2 = rb.state_jintegrate(model, x, dx)
            ddx_dx = np.zeros((ndx, ndx)); ddx_dx[:nv, nv:] = dt * np.eye(nv)
            out["Fx"] = J1 + J2 @ ddx_dx
            out["Fu"] = J2 @ np.vstack([dt * dt * np.eye(nv), dt * np.eye(nv)])
            out.update(Lx=dt * Lx, Lxx=dt * Lxx, Lu=dt * Lu, Luu=dt * Luu)
    elif diff:
        out.update(Lx=Lx, Lxx=Lxx)
    return out


# ------------------------------------------------------------------------ SolverDDP ---
def solve_ddp(prob, x0, maxiter=100, reg_min=1e-9, reg_max=1e9, verbose=False):
    """crocoddyl 1.9.0 SolverDDP::solve(init_xs={}, init_us={}, maxiter=100, is_feasible=false,
    reginit=NaN) as called by InverseKinematics::optimize (inverse_kinematics.cpp:56-58).
    Returns dict(xs, us, iters, cost, stop, converged, reg)."""
    model, T = prob.model, prob.T
    nv = model.nv
    ndx = 2 * nv
    zero = np.concatenate([rb.neutral(model), np.zeros(nv)])
    xs = [zero.copy() for _ in range(T + 1)]
    us = [np.zeros(nv) for _ in range(T)]
    xs_try = [np.array(x0, float)] + [None] * T
    us_try = [None] * T
    is_feasible, was_feasible = False, False
    xreg = ureg = reg_min
    th_grad, th_gaptol, th_stepdec, th_stepinc, th_accept, th_stop = 1e-12, 1e-16, 0.5, 0.01, 0.1, 1e-9
    alphas = [2.0 ** (-k) for k in range(10)]
    cost = 0.0
    fs = [np.zeros(ndx) for _ in range(T + 1)]
    data = [None] * (T + 1)
    K = [None] * T; k = [None] * T; Qu = [None] * T; Quuk = [None] * T
    stop, it_done, converged = np.inf, 0, False

    def calc_diff():
        nonlocal cost, is_feasible
        cost = 0.0
        for t in range(T):
            data[t] = node_calc(prob, t, xs[t], us[t], diff=True)
            cost += data[t]["cost"]
        data[T] = node_calc(prob, T, xs[T], None, diff=True)
        cost += data[T]["cost"]
        if not is_feasible:
            fs[0] = rb.state_diff(model, xs[0], x0)
            ok = np.max(np.abs(fs[0])) < th_gaptol
            for t in range(T):
                fs[t + 1] = rb.state_diff(model, xs[t + 1], data[t]["xnext"])
                ok = ok and np.max(np.abs(fs[t + 1])) < th_gaptol
            is_feasible = ok
        elif not was_feasible:
            for t in range(T + 1):
                fs[t] = np.zeros(ndx)

    def backward():
        Vxx = data[T]["Lxx"].copy()
        Vx = data[T]["Lx"].copy()
        Vxx[np.diag_indices(ndx)] += xreg
        if not is_feasible:
            Vx = Vx + Vxx @ fs[T]
        for t in range(T - 1, -1, -1):
            d = data[t]
            FxTV = d["Fx"].T @ Vxx
            Qxx = d["Lxx"] + FxTV @ d["Fx"]
            Qx = d["Lx"] + d["Fx"].T @ Vx
            FuTV = d["Fu"].T @ Vxx
            Qxu = FxTV @ d["Fu"]
            Quu = d["Luu"] + FuTV @ d["Fu"]
            Qu[t] = d["Lu"] + d["Fu"].T @ Vx
            Quu[np.diag_indices(nv)] += ureg
            L = np.linalg.cholesky(Quu)          # raises LinAlgError when not positive definite
            K[t] = np.linalg.solve(L.T, np.linalg.solve(L, Qxu.T))
            k[t] = np.linalg.solve(L.T, np.linalg.solve(L, Qu[t]))
            Quuk[t] = Quu @ k[t]
            Vx = Qx - K[t].T @ Qu[t]
            Vxx = Qxx - Qxu @ K[t]
            Vxx = 0.5 * (Vxx + Vxx.T)
            Vxx[np.diag_indices(ndx)] += xreg
            if not is_feasible:
                Vx = Vx + Vxx @ fs[t]
            if not (np.isfinite(Vx).all() and np.isfinite(Vxx).all()):
                raise FloatingPointError("backward_error")

    def forward(alpha):
        c = 0.0
        for t in range(T):
            dx = rb.state_diff(model, xs[t], xs_try[t])
            us_try[t] = us[t] - alpha * k[t] - K[t] @ dx
            d = node_calc(prob, t, xs_try[t], us_try[t])
            xs_try[t + 1] = d["xnext"]
            c += d["cost"]
            if not (np.isfinite(c) and np.isfinite(xs_try[t + 1]).all()):
                raise FloatingPointError("forward_error")
        c += node_calc(prob, T, xs_try[T], None)["cost"]
        if not np.isfinite(c):
            raise FloatingPointError("forward_error")
        return c

    recalc = True
    for it in range(maxiter):
        it_done = it + 1
        while True:
            try:
                if recalc:
                    calc_diff()
                backward()
            except (np.linalg.LinAlgError, FloatingPointError):
                recalc = False
                xreg = ureg = min(xreg * 10.0, reg_max)
                if xreg == reg_max:
                    return dict(xs=xs, us=us, iters=it_done, cost=cost, stop=stop, converged=False, reg=xreg)
                continue
            break
        d1 = sum(Qu[t] @ k[t] for t in range(T))
        d2 = -sum(k[t] @ Quuk[t] for t in range(T))
        recalc = False
        alpha = alphas[-1]
        for alpha in alphas:
            try:
                cost_try = forward(alpha)
            except FloatingPointError:
                continue
            dV = cost - cost_try
            dVexp = alpha * (d1 + 0.5 * alpha * d2)
            if dVexp >= 0:
                if d1 < th_grad or not is_feasible or dV > th_accept * dVexp:
                    was_feasible = is_feasible
                    xs = [a.copy() for a in xs_try]
                    us = [a.copy() for a in us_try]
                    is_feasible = True
                    cost = cost_try
                    recalc = True
                    break
        if alpha > th_stepdec:
            xreg = ureg = max(xreg / 10.0, reg_min)
        if alpha <= th_stepinc:
            xreg = ureg = min(xreg * 10.0, reg_max)
            if xreg == reg_max:
                return dict(xs=xs, us=us, iters=it_done, cost=cost, stop=stop, converged=False, reg=xreg)
        stop = sum(Qu[t] @ Qu[t] for t in range(T))
        if verbose:
            print("iter %3d cost %.6e stop %.3e alpha %.4f reg %.1e feas %d" % (it, cost, stop, alpha, xreg, is_feasible))
        if was_feasible and stop < th_stop:
            converged = True
            break
    return dict(xs=xs, us=us, iters=it_done, cost=cost, stop=stop, converged=converged, reg=xreg)

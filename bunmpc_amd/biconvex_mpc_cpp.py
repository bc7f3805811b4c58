"""Drop-in for the reference's pybind module `biconvex_mpc_cpp`
(iterative_supervised_learning/srcpy/motion_planner/biconvex.cpp:15-66): `BiconvexMP`
(same method names, argument meaning and print-and-continue error behaviour), backed by
the C-ABI in include/bunmpc.h; `optimize` runs the gfx950 ADMM kernel.

Differences that are deliberate and visible: `set_cost_x/f` accept only a diagonal Q (the
only form the reference's own callers build); wrong sizes raise ValueError instead of
reading out of bounds."""
import ctypes as C

import numpy as np

from . import _lib


def _vec(a, n, name):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if a.shape[0] != n:
        raise ValueError("%s: expected %d values, got %d" % (name, n, a.shape[0]))
    return a


def _diag_of(Q, n, name):
    if hasattr(Q, "tocoo"):  # scipy sparse
        coo = Q.tocoo()
        if coo.shape != (n, n):
            raise ValueError("%s: expected a %dx%d matrix" % (name, n, n))
        off = coo.row != coo.col
        if np.any(coo.data[off] != 0):
            raise ValueError("%s: only diagonal Q is supported" % name)
        d = np.zeros(n)
        np.add.at(d, coo.row[~off], coo.data[~off])
        return d
    Q = np.asarray(Q, dtype=np.float64)
    if Q.ndim == 1:
        return _vec(Q, n, name)
    if Q.shape != (n, n):
        raise ValueError("%s: expected a %dx%d matrix" % (name, n, n))
    if np.any(Q - np.diag(np.diag(Q)) != 0):
        raise ValueError("%s: only diagonal Q is supported" % name)
    return np.ascontiguousarray(np.diag(Q))


class BiconvexMP:
    """motion_planner::BiConvexMP (biconvex.hpp:21-192)."""

    def __init__(self, m, n_col, n_eff, _handle=None, _owner=None):
        self._lib = _lib.lib()
        self._owner = _owner   # keeps a KinoDynMP alive when this object is its `dyn`
        self._owned = _handle is None
        self._h = self._lib.bmpc_biconvex_create(float(m), int(n_col), int(n_eff)) if _handle is None else _handle
        if not self._h:
            raise _lib.BmpcError(_lib.BAD_ARG, _lib.last_error())
        self.n_col, self.n_eff = int(n_col), int(n_eff)
        self.nx, self.nf = 9 * (self.n_col + 1), 3 * self.n_eff * self.n_col

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self, "_owned", False):
            self._lib.bmpc_biconvex_destroy(h)

    def set_contact_plan(self, cnt_plan, dt):
        c = np.ascontiguousarray(cnt_plan, dtype=np.float64)
        if c.shape != (self.n_eff, 4):
            raise ValueError("cnt_plan must be (n_eff, 4)")
        _lib.check(self._lib.bmpc_biconvex_set_contact_plan(self._h, c.ctypes.data, float(dt)))

    def set_rotation_matrix_f(self, rot_matrix):
        R = _vec(rot_matrix, 9, "rot_matrix")
        _lib.check(self._lib.bmpc_biconvex_set_rotation_matrix_f(self._h, R.ctypes.data))

    def return_A_x(self, X):
        out = np.zeros((self.nx, self.nf))
        _lib.check(self._lib.bmpc_biconvex_return_A_x(self._h, _vec(X, self.nx, "X").ctypes.data, out.ctypes.data))
        return out

    def return_b_x(self, X):
        out = np.zeros(self.nx)
        _lib.check(self._lib.bmpc_biconvex_return_b_x(self._h, _vec(X, self.nx, "X").ctypes.data, out.ctypes.data))
        return out

    def return_A_f(self, F, x_init):
        out = np.zeros((self.nx, self.nx))
        _lib.check(self._lib.bmpc_biconvex_return_A_f(self._h, _vec(F, self.nf, "F").ctypes.data,
                                                      _vec(x_init, 9, "x_init").ctypes.data, out.ctypes.data))
        return out

    def return_b_f(self, F, x_init):
        out = np.zeros(self.nx)
        _lib.check(self._lib.bmpc_biconvex_return_b_f(self._h, _vec(F, self.nf, "F").ctypes.data,
                                                      _vec(x_init, 9, "x_init").ctypes.data, out.ctypes.data))
        return out

    def set_cost_x(self, Q_x, q_x):
        Q, q = _diag_of(Q_x, self.nx, "Q_x"), _vec(q_x, self.nx, "q_x")
        _lib.check(self._lib.bmpc_biconvex_set_cost_x(self._h, Q.ctypes.data, q.ctypes.data))

    def set_cost_f(self, Q_f, q_f):
        Q, q = _diag_of(Q_f, self.nf, "Q_f"), _vec(q_f, self.nf, "q_f")
        _lib.check(self._lib.bmpc_biconvex_set_cost_f(self._h, Q.ctypes.data, q.ctypes.data))

    def create_cost_X(self, W_X, W_X_ter, X_ter, X_nom):
        a = [_vec(W_X, self.nx - 9, "W_X"), _vec(W_X_ter, 9, "W_X_ter"), _vec(X_ter, 9, "X_ter"),
             _vec(X_nom, self.nx - 9, "X_nom")]
        _lib.check(self._lib.bmpc_biconvex_create_cost_X(self._h, *[x.ctypes.data for x in a]))

    def create_cost_F(self, W_F):
        w = _vec(W_F, self.nf, "W_F")
        _lib.check(self._lib.bmpc_biconvex_create_cost_F(self._h, w.ctypes.data))

    def set_bounds_x(self, lb, ub):
        lb, ub = _vec(lb, self.nx, "lb"), _vec(ub, self.nx, "ub")
        _lib.check(self._lib.bmpc_biconvex_set_bounds_x(self._h, lb.ctypes.data, ub.ctypes.data))

    def set_bounds_f(self, lb, ub):
        lb, ub = _vec(lb, self.nf, "lb"), _vec(ub, self.nf, "ub")
        _lib.check(self._lib.bmpc_biconvex_set_bounds_f(self._h, lb.ctypes.data, ub.ctypes.data))

    def create_bound_constraints(self, b, fx_max, fy_max, fz_max):
        b = np.ascontiguousarray(b, dtype=np.float64)
        if b.ndim != 2:
            raise ValueError("b must be a matrix")
        _lib.check(self._lib.bmpc_biconvex_create_bound_constraints(
            self._h, b.ctypes.data, b.shape[0], b.shape[1], float(fx_max), float(fy_max), float(fz_max)))

    def set_rho(self, rho):
        _lib.check(self._lib.bmpc_biconvex_set_rho(self._h, float(rho)))

    def _out(self, fn, shape):
        out = np.zeros(shape)
        _lib.check(fn(self._h, out.ctypes.data))
        return out

    def return_opt_x(self):
        return self._out(self._lib.bmpc_biconvex_return_opt_x, self.nx)

    def return_opt_f(self):
        return self._out(self._lib.bmpc_biconvex_return_opt_f, self.nf)

    def return_opt_p(self):
        return self._out(self._lib.bmpc_biconvex_return_opt_p, self.nx)

    def return_opt_com(self):
        return self._out(self._lib.bmpc_biconvex_return_opt_com, (self.n_col + 1, 3))

    def return_opt_mom(self):
        return self._out(self._lib.bmpc_biconvex_return_opt_mom, (self.n_col + 1, 6))

    def set_warm_start_vars(self, x_wm, f_wm, P_wm):
        a = [_vec(x_wm, self.nx, "x_wm"), _vec(f_wm, self.nf, "f_wm"), _vec(P_wm, self.nx, "P_wm")]
        _lib.check(self._lib.bmpc_biconvex_set_warm_start_vars(self._h, *[x.ctypes.data for x in a]))

    def optimize(self, x_init, num_iters):
        """Like the reference: returns None; on divergence the C side prints
        "ERROR: solver diverged, Dyn violation is NaN" and the iterates hold NaNs."""
        x = _vec(x_init, 9, "x_init")
        rc = self._lib.bmpc_biconvex_optimize(self._h, x.ctypes.data, int(num_iters))
        if rc not in (_lib.OK, _lib.DIVERGED):
            _lib.check(rc)

    def return_dyn_viol_hist(self):
        n = self._lib.bmpc_biconvex_dyn_viol_hist_size(self._h)
        out = np.zeros(max(n, 1))
        _lib.check(self._lib.bmpc_biconvex_return_dyn_viol_hist(self._h, out.ctypes.data))
        return [float(v) for v in out[:n]]

    def collect_statistics(self):
        _lib.check(self._lib.bmpc_biconvex_collect_statistics(self._h))

    # additive (not in the reference binding)
    def set_friction_coefficient(self, mu):
        _lib.check(self._lib.bmpc_biconvex_set_friction_coefficient(self._h, float(mu)))

    def set_robot_mass(self, m):
        _lib.check(self._lib.bmpc_biconvex_set_robot_mass(self._h, float(m)))

    def step_constants(self):
        a, b = C.c_double(), C.c_double()
        _lib.check(self._lib.bmpc_biconvex_get_step_constants(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_step_constants(self, L_x, L_f):
        _lib.check(self._lib.bmpc_biconvex_set_step_constants(self._h, float(L_x), float(L_f)))

    def last_stats(self):
        s = np.zeros(_lib.NSTATS, dtype=np.int32)
        _lib.check(self._lib.bmpc_biconvex_last_stats(self._h, s.ctypes.data))
        return s.astype(np.int64)


class CentroidalDynamics:
    """Bound with its constructor only (srcpy/motion_planner/biconvex.cpp:51-52)."""

    def __init__(self, m, n_col, n_eff):
        self.m, self.n_col, self.n_eff = float(m), int(n_col), int(n_eff)


class KinoDynMP:
    """motion_planner::KinoDynMP (srcpy/motion_planner/biconvex.cpp:55-63, src/motion_planner/kino_dyn.cpp).
    `urdf` may be a URDF path (as in the reference), a RobotModel or a prepared DeviceModel."""

    def __init__(self, urdf, m, n_eff, dyn_col, ik_col):
        from .inverse_kinematics_cpp import InverseKinematics, as_device_model
        self._lib = _lib.lib()
        self._dm = as_device_model(urdf)
        self._h = self._lib.bmpc_kinodyn_create(self._dm.h, float(m), int(n_eff), int(dyn_col), int(ik_col))
        if not self._h:
            raise _lib.BmpcError(_lib.BAD_ARG, _lib.last_error())
        # return_dyn / return_ik hand out references to members (no keep_alive in the reference):
        # here the wrappers hold a reference back to this object instead
        self._dyn = BiconvexMP(m, dyn_col, n_eff, _handle=self._lib.bmpc_kinodyn_return_dyn(self._h), _owner=self)
        self._ik = InverseKinematics(None, ik_col, _handle=self._lib.bmpc_kinodyn_return_ik(self._h), _owner=self,
                                     _dmodel=self._dm)
        self.nq, self.nv = self._dm.model.nq, self._dm.model.nv

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.bmpc_kinodyn_destroy(h)

    def return_dyn(self):
        return self._dyn

    def return_ik(self):
        return self._ik

    def optimize(self, q, v, dyn_iters, kino_dyn_iters):
        q, v = _vec(q, self.nq, "q"), _vec(v, self.nv, "v")
        rc = self._lib.bmpc_kinodyn_optimize(self._h, q.ctypes.data, v.ctypes.data, int(dyn_iters), int(kino_dyn_iters))
        if rc not in (_lib.OK, _lib.DIVERGED):
            _lib.check(rc)

    def set_com_tracking_weight(self, wt_com):     # the harness passes a 1-element ndarray (abstract_cyclic_gen.py:136-141)
        _lib.check(self._lib.bmpc_kinodyn_set_com_tracking_weight(self._h, float(np.asarray(wt_com).reshape(-1)[0])))

    def set_mom_tracking_weight(self, wt_mom):
        _lib.check(self._lib.bmpc_kinodyn_set_mom_tracking_weight(self._h, float(np.asarray(wt_mom).reshape(-1)[0])))

    def compute_solve_times(self):
        _lib.check(self._lib.bmpc_kinodyn_compute_solve_times(self._h))

    def return_solve_times(self):
        out = np.zeros(3)
        _lib.check(self._lib.bmpc_kinodyn_return_solve_times(self._h, out.ctypes.data))
        return out

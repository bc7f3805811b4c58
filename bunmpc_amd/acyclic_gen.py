"""SoloAcyclicGen: the acyclic-motion harness (jumps, rearing, cartwheels)
(`ISL/examples/mpc/abstract_acyclic_gen.py:13-369`, ISL = iterative_supervised_learning) on this package's solver classes.
Same methods and the same time-table look-ups: a plan object (`motions/weight_abstract.py::ACyclicMotionParams`, the
reference's own `motions/acyclic/*.py` files work as they are) lists contact phases, nominal centroidal states, bounds, swing
targets and state / control regularisation -- each with a [start, end) time -- and every MPC call slices them at the knot
times.  The regularisation reference and weights therefore change from node to node, which the IK kernels take as per-node
vectors (`sn_*` strides of bmpc_ik_batch_t).  Plot / save helpers are not carried over.
"""
import numpy as np

from . import fk_np
from .biconvex_mpc_cpp import KinoDynMP
from .urdf_model import RobotModel, load_urdf


class SoloAcyclicGen:
    eff_names = ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"]        # :27

    def __init__(self, robot, r_urdf, eff_names=None):
        if robot is None:
            robot = r_urdf if isinstance(r_urdf, RobotModel) else load_urdf(r_urdf)
        self.rmodel = robot
        self.r_urdf = r_urdf if r_urdf is not None else robot
        self.m = robot.total_mass
        if eff_names is not None:
            self.eff_names = list(eff_names)
        self.n_eff = 4
        self.ee_frame_id = [robot.frame_id(n) for n in self.eff_names]
        self.fx_max = self.fy_max = self.fz_max = 25.0                # :34-36
        self.dyn_iters = 50                                           # kd.optimize(q, v, 50, 1)  (:318)

    def _make_kd(self):
        return KinoDynMP(self.r_urdf, self.m, len(self.eff_names), self.horizon, self.ik_horizon)

    def update_motion_params(self, weight_abstract, q0, t0):
        """:42-72"""
        self.q0, self.t0 = q0, t0
        p = self.params = weight_abstract
        self.freq = p.plan_freq[0][0]
        self.horizon = self.ik_horizon = p.n_col
        self.kd = self._make_kd()
        self.kd.set_com_tracking_weight(p.cent_wt[0])
        self.kd.set_mom_tracking_weight(p.cent_wt[1])
        self.ik = self.kd.return_ik()
        self.mp = self.kd.return_dyn()
        self.mp.set_rho(p.rho)
        self.size = min(self.ik_horizon, int(self.freq / p.dt_arr[0]) + 2)
        if self.freq > p.dt_arr[0]:
            self.size += 1

    def _knot_times(self, t, n, clamp):
        """ft after the k-th increment, k = 0..n-1, as the reference accumulates it (round(., 3) after every step)"""
        p = self.params
        ft = t - p.dt_arr[0] - self.t0
        out = []
        for i in range(n):
            ft = np.round(ft + p.dt_arr[min(i, clamp)], 3)
            out.append(ft)
        return out

    def create_contact_plan(self, q, v, t, make_cyclic=False):
        """:74-122 (the cyclic continuation is a `pass` there too)"""
        p = self.params
        self.cnt_plan = np.zeros((self.horizon, len(self.eff_names), 4))
        ft = np.round(t - p.dt_arr[0] - self.t0, 3)
        for i in range(p.n_col):
            ft += np.round(p.dt_arr[i], 3)
            if ft < p.cnt_plan[-1][0][5]:
                for k in range(len(p.cnt_plan)):
                    if p.cnt_plan[k][0][4] <= ft < p.cnt_plan[k][0][5]:
                        for j in range(len(self.eff_names)):
                            self.cnt_plan[i][j] = p.cnt_plan[k][j][0:4]
                        break
            elif not make_cyclic:
                for j in range(len(self.eff_names)):
                    self.cnt_plan[i][j] = p.cnt_plan[-1][j][0:4]
            if i == 0:
                dt = p.dt_arr[i] - np.round(np.remainder(t, p.dt_arr[i]), 2)
                if dt == 0:
                    dt = p.dt_arr[i]
            else:
                dt = p.dt_arr[i]
            self.mp.set_contact_plan(self.cnt_plan[i], dt)

    def create_costs(self, q, v, t, make_cyclic=False):
        """:124-296"""
        p = self.params
        q, v = np.asarray(q, float), np.asarray(v, float)
        nq, nv = self.rmodel.nq, self.rmodel.nv
        self.x0 = np.hstack((q, v))
        kin = fk_np.kinematics(self.rmodel, q[None], v[None])
        X_init = np.concatenate([kin["com"][0], kin["vcom"][0], kin["L"][0]])
        # --- dynamics costs
        X_nom = np.zeros(9 * self.horizon)
        X_ter = None
        times = self._knot_times(t, p.n_col, p.n_col - 1)
        for i, ft in enumerate(times):
            if ft < p.X_nom[-1][-1]:
                for k in range(len(p.X_nom)):
                    if p.X_nom[k][9] <= ft < p.X_nom[k][10]:
                        X_nom[9 * i:9 * (i + 1)] = p.X_nom[k][0:9]
                        break
                if i == p.n_col - 1:
                    X_ter = X_nom[-9:].copy()
            elif not make_cyclic:
                X_nom[9 * i:9 * (i + 1)] = p.X_ter
                if i == p.n_col - 1:
                    X_ter = np.asarray(p.X_ter, float)
        self.bounds = np.zeros((self.horizon, 6))
        for i, ft in enumerate(times):
            if ft < p.bounds[-1][-1]:
                for k in range(len(p.bounds)):
                    if p.bounds[k][-2] <= ft < p.bounds[k][-1]:
                        self.bounds[i] = p.bounds[k][0:6]
                        break
            elif not make_cyclic:
                self.bounds[i] = p.bounds[-1][0:6]
        X_nom[0:9] = X_init
        self.X_nom, self.X_ter = X_nom, X_ter
        self.mp.create_bound_constraints(self.bounds, self.fx_max, self.fy_max, self.fz_max)
        self.mp.create_cost_X(np.tile(p.W_X, self.horizon), p.W_X_ter, X_ter, X_nom)
        self.mp.create_cost_F(np.tile(p.W_F, self.horizon))
        # --- IK costs
        T = self.ik_horizon
        self.dt_arr = np.zeros(T + 1)
        for i in range(T):
            for j in range(len(self.eff_names)):
                if self.cnt_plan[i][j][0] == 1:
                    self.ik.add_position_tracking_task_single(self.ee_frame_id[j], self.cnt_plan[i][j][1:4], p.cnt_wt,
                                                              "cnt_0" + self.eff_names[j], i)
        if isinstance(p.swing_wt, (np.ndarray, list)):
            for i, ft in enumerate(self._knot_times(t, T, T - 1)):
                if ft < p.swing_wt[-1][0][5]:
                    for k in range(len(p.swing_wt)):
                        if p.swing_wt[k][0][4] <= ft < p.swing_wt[k][0][5]:
                            for j in range(len(self.eff_names)):
                                if p.swing_wt[k][j][0] > 0:
                                    self.ik.add_position_tracking_task_single(self.ee_frame_id[j], p.swing_wt[k][j][1:4],
                                                                              p.swing_wt[k][j][0], "swing_0" + self.eff_names[j], i)
                            break
        # state regularisation, node by node (terminal = node T)
        for i, ft in enumerate(self._knot_times(t, T + 1, T - 1)):
            self.dt_arr[min(i, T - 1)] = p.dt_arr[min(i, T - 1)]
            k_sel = len(p.state_reg) - 1 if not make_cyclic else None
            if ft < p.state_reg[-1][-1]:
                k_sel = None
                for k in range(len(p.state_reg)):
                    if p.state_scale[k][1] <= ft < p.state_scale[k][2]:
                        k_sel = k
                        break
            if k_sel is None:
                continue
            wts, ref = np.asarray(p.state_wt[k_sel][0:2 * nv], float), np.asarray(p.state_reg[k_sel][0:nq + nv], float)
            if i < p.n_col:
                self.ik.add_state_regularization_cost_single(i, p.state_scale[k_sel][0], "xReg", wts, ref)
            else:
                self.ik.add_state_regularization_cost(0, i, p.state_scale[k_sel][0], "xReg", wts, ref, True)
        # control regularisation (note the reference passes ctrl_wt[k][0] as the weight of the running nodes, :262,277)
        for i, ft in enumerate(self._knot_times(t, p.n_col + 1, p.n_col - 1)):
            k_sel = len(p.ctrl_scale) - 1 if not make_cyclic else None
            if ft < p.ctrl_scale[-1][-1]:
                k_sel = None
                for k in range(len(p.ctrl_scale)):
                    if p.ctrl_scale[k][1] <= ft < p.ctrl_scale[k][2]:
                        k_sel = k
                        break
            if k_sel is None:
                continue
            wts, ref = np.asarray(p.ctrl_wt[k_sel][0:nv], float), np.asarray(p.ctrl_reg[k_sel][0:nv], float)
            if i < p.n_col:
                self.ik.add_ctrl_regularization_cost_single(i, p.ctrl_wt[k_sel][0], "ctrlReg", wts, ref)
            else:
                self.ik.add_ctrl_regularization_cost(0, i, p.ctrl_scale[k_sel][0], "ctrlReg", wts, ref, True)
        self.ik.setup_costs(self.dt_arr)

    def optimize(self, q, v, t, X_wm=None, F_wm=None, P_wm=None):
        """:298-347: plan, solve, and the zero-order-hold 1 kHz expansion (linspace(x_i, x_i, int(dt_i / 0.001)))"""
        self.create_contact_plan(q, v, t)
        self.create_costs(q, v, t)
        self.kd.optimize(np.asarray(q, float), np.asarray(v, float), self.dyn_iters, 1)
        F = self.mp.return_opt_f().reshape(self.horizon, 3 * len(self.eff_names))
        xs, us = np.array(self.ik.get_xs()), np.array(self.ik.get_us())
        n = [int(self.dt_arr[i] / 0.001) for i in range(len(xs))]
        self.xs_int = np.vstack([np.tile(xs[i], (n[i], 1)) for i in range(len(xs))])
        self.us_int = np.vstack([np.tile(us[i], (n[i], 1)) for i in range(len(xs) - 1)])
        self.f_int = np.vstack([np.tile(F[i], (n[i], 1)) for i in range(len(xs) - 1)])
        return self.xs_int, self.us_int, self.f_int

    def get_plan_freq(self, t):
        """:349-358"""
        pf = self.params.plan_freq
        for k in range(len(pf)):
            if t - self.t0 < pf[-1][-1]:
                if pf[k][-2] <= t - self.t0 < pf[k][-1]:
                    return pf[k][0]
            else:
                return pf[-1][0]

    def get_gains(self, t):
        """:360-369"""
        kp, kd = self.params.kp, self.params.kd
        for k in range(len(kp)):
            if t - self.t0 < kp[-1][-1]:
                if kp[k][-2] <= t - self.t0 < kp[k][-1]:
                    return kp[k][0], kd[k][0]
            else:
                return kp[-1][0], kd[-1][0]

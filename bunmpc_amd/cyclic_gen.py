"""SoloMpcGaitGen: the cyclic-gait MPC harness the rollout scripts drive
(`ISL/examples/mpc/abstract_cyclic_gen.py:15-698`, ISL = iterative_supervised_learning), rebuilt on
this package: kinematics from a RobotModel (fk_np) instead of pinocchio, the solve on the GPU through
KinoDynMP (C-ABI).  Same method names, arguments and return values for the data path
    update_gait_params -> optimize(q, v, t, v_des, w_des) -> (xs_int, us_int, f_int)
Not carried over: MCTS contact locations, contact-location noise, height maps, `create_cnt_plan_bis`
and the matplotlib helpers (they are outside the solve path; passing them raises NotImplementedError).

The contact plan and cost builders are the batch functions of `problems.py` called with B = 1, so a
single MPC call here and one row of a batch are the same arithmetic.
"""
import numpy as np

from . import fk_np
from .biconvex_mpc_cpp import KinoDynMP
from .gait_planner_cpp import GaitPlanner
from .problems import BOUNDS_TILE, FOOT_SIZE, RobotParams, _log3_batch, centroidal_costs, contact_plan
from .urdf_model import RobotModel, load_urdf


def yaw_only(R):
    """matrixToRpy -> zero roll and pitch -> rpyToMatrix (abstract_cyclic_gen.py:173-177,588-591)"""
    y = np.arctan2(R[1, 0], R[0, 0])
    c, s = np.cos(y), np.sin(y)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def composite_inertia_base(model, q):
    """rdata.Ycrb[1].inertia after crba(q) (:48-49): rotational inertia of the whole robot about its
    CoM, expressed in the base joint frame"""
    k = fk_np.kinematics(model, np.asarray(q, float)[None])
    I = np.zeros((3, 3))
    for b in range(model.nj + 1):
        R = k["oR"][b][0]
        d = R @ model.com[b] + k["op"][b][0] - k["com"][0]
        I += R @ model.inertia[b] @ R.T + model.mass[b] * ((d @ d) * np.eye(3) - np.outer(d, d))
    Rb = k["oR"][0][0]
    return Rb.T @ I @ Rb


def interpolate_plan(knots, dt_arr, size, step=0.001):
    """the 1 kHz resampling of the first `size` knot intervals (:677-692):
    vstack_i linspace(knots[i], knots[i+1], int(dt_i / step)) -- end points included, as there"""
    return np.vstack([np.linspace(knots[i], knots[i + 1], int(dt_arr[i] / step)) for i in range(size)])


class SoloMpcGaitGen:
    eff_names = ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"]     # :37
    hip_names = ["FL_HFE", "FR_HFE", "HL_HFE", "HR_HFE"]         # :38

    def __init__(self, robot, r_urdf, x_reg, planning_time, q0, height_map=None, eff_names=None, hip_names=None):
        """robot: RobotModel, or None to build it from r_urdf (the reference passes a pinocchio
        RobotWrapper here); r_urdf: URDF path or RobotModel handed on to KinoDynMP."""
        if height_map is not None:
            raise NotImplementedError("height maps are outside the solve path")
        if robot is None:
            robot = r_urdf if isinstance(r_urdf, RobotModel) else load_urdf(r_urdf)
        self.rmodel = robot
        self.r_urdf = r_urdf if r_urdf is not None else robot
        self.foot_size = FOOT_SIZE
        self.planning_time = planning_time
        if eff_names is not None:
            self.eff_names = list(eff_names)
        if hip_names is not None:
            self.hip_names = list(hip_names)
        self.n_eff = 4
        q0 = np.asarray(q0, float)
        k0 = fk_np.kinematics(robot, q0[None])
        self.I_composite_b = composite_inertia_base(robot, q0)
        self.gravity = 9.81
        self.ee_frame_id = [robot.frame_id(n) for n in self.eff_names]
        off = np.round(fk_np.frame_positions(robot, k0, self.hip_names)[0] - k0["com"][0], 3)       # :56-59
        off[:, 1] += np.array([0.04, -0.04, 0.04, -0.04])                                          # :61-72
        self.offsets = off @ k0["oR"][0][0]                                                        # R^T off_i  (:76-79)
        self.x_reg = np.asarray(x_reg, float)
        self.m = robot.total_mass
        self.bx = self.by = self.bz = 0.45
        self.fx_max = self.fy_max = self.fz_max = 15.0
        self.height_map = None
        self.q_traj, self.v_traj, self.xs_traj = [], [], []

    def update_gait_params(self, weight_abstract, t, ik_hor_ratio=0.5, horizon=None):
        """:108-156"""
        p = self.params = weight_abstract
        self.gait_planner = GaitPlanner(p.gait_period, np.array(p.stance_percent), np.array(p.phase_offset), p.step_ht)
        self.gait_horizon = p.gait_horizon
        self.horizon = horizon if horizon is not None else int(np.round(p.gait_horizon * p.gait_period / p.gait_dt, 2))
        self.ik_horizon = int(np.round(ik_hor_ratio * p.gait_horizon * p.gait_period / p.gait_dt, 2))
        self.dt_arr = np.zeros(self.horizon)
        self.kd = KinoDynMP(self.r_urdf, self.m, len(self.eff_names), self.horizon, self.ik_horizon)
        self.kd.set_com_tracking_weight(np.atleast_1d(p.cent_wt[0]))
        self.kd.set_mom_tracking_weight(np.atleast_1d(p.cent_wt[1]))
        self.ik = self.kd.return_ik()
        self.mp = self.kd.return_dyn()
        self.mp.set_rho(p.rho)
        self.X_nom = np.zeros(9 * self.horizon)
        self.size = min(self.ik_horizon, int(self.planning_time / p.gait_dt) + 2)
        if self.planning_time > p.gait_dt:
            self.size -= 1

    def _robot_params(self, Ryaw):
        hip = (self.offsets @ Ryaw.T)[:, 0:2]                       # (R offsets[j])[0:2]
        return hip

    def create_cnt_plan(self, q, v, t, v_des, w_des, noise_std=None, mcts_x_y_cnt_loc=None, ee_pos=None, z_height=None):
        """:159-414, data path"""
        if noise_std is not None or mcts_x_y_cnt_loc is not None:
            raise NotImplementedError("contact-location noise / MCTS locations are outside the solve path")
        p = self.params
        q, v = np.asarray(q, float), np.asarray(v, float)
        kin = fk_np.kinematics(self.rmodel, q[None], v[None])
        if z_height is None:
            com, zh = np.round(kin["com"][0, 0:2], 3), kin["com"][0, 2]
        else:
            com, zh = np.round(np.asarray(z_height)[0:2], 3), z_height[2]
        feet0 = (np.round(fk_np.frame_positions(self.rmodel, kin, self.eff_names)[0], 3) if ee_pos is None
                 else np.asarray(ee_pos, float))
        hip = self._robot_params(yaw_only(kin["oR"][0][0]))
        rp = RobotParams("robot", self.m, feet0[:, 0:2], hip, zh)
        v_des = np.asarray(v_des, float)
        cnt, swing, dt = contact_plan(p, rp, self.horizon, np.array([float(t)]), com[None], np.array([zh]),
                                      feet0[None], v_des[None], np.array([float(w_des)]), hip[None])
        self.cnt_plan, self.swing_time, self.dt_arr = cnt[0], swing[0], dt[0]
        for i in range(self.horizon):
            self.mp.set_contact_plan(self.cnt_plan[i], self.dt_arr[i])
        return self.cnt_plan

    def compute_ori_correction(self, q, des_R):
        """log3(R_des R_q^T) (:616-627)"""
        Rq = fk_np._quat_R(np.asarray(q, float)[None, 3:7])[0]
        return _log3_batch((des_R @ Rq.T)[None])[0]

    def create_costs(self, q, v, v_des, w_des, ori_des):
        """:532-614"""
        p = self.params
        q, v, v_des = np.asarray(q, float), np.asarray(v, float), np.asarray(v_des, float)
        self.x0 = np.hstack((q, v))
        T = self.ik_horizon
        for i in range(T):
            for j in range(len(self.eff_names)):
                if self.cnt_plan[i][j][0] == 1:
                    self.ik.add_position_tracking_task_single(self.ee_frame_id[j], self.cnt_plan[i][j][1:4], p.swing_wt[0],
                                                              "cnt_0" + self.eff_names[j], i)
                elif self.swing_time[i][j] == 1:
                    pos = self.cnt_plan[i][j][1:4].copy()
                    pos[2] = p.step_ht
                    self.ik.add_position_tracking_task_single(self.ee_frame_id[j], pos, p.swing_wt[1],
                                                              "via_0" + self.eff_names[j], i)
        nv = self.rmodel.nv
        self.ik.add_state_regularization_cost(0, T, p.reg_wt[0], "xReg", np.asarray(p.state_wt, float), self.x_reg, False)
        self.ik.add_ctrl_regularization_cost(0, T, p.reg_wt[1], "uReg", np.asarray(p.ctrl_wt, float), np.zeros(nv), False)
        self.ik.add_state_regularization_cost(0, T, p.reg_wt[0], "xReg", np.asarray(p.state_wt, float), self.x_reg, True)
        self.ik.add_ctrl_regularization_cost(0, T, p.reg_wt[1], "uReg", np.asarray(p.ctrl_wt, float), np.zeros(nv), True)
        self.ik.setup_costs(self.dt_arr[0:T])

        kin = fk_np.kinematics(self.rmodel, q[None], v[None])
        self.X_init = np.concatenate([kin["com"][0], kin["vcom"][0], kin["L"][0]])
        des_R = yaw_only(fk_np._quat_R(np.asarray(ori_des, float)[None])[0])                       # :588-591
        amom = self.compute_ori_correction(q, des_R)
        X_nom, X_ter = centroidal_costs(p, self.horizon, self.X_init[None], v_des[None], self.dt_arr[None], amom[None])
        self.X_nom, X_ter = X_nom[0], X_ter[0]
        if w_des != 0:                                                                             # :603-608
            yaw_momentum = (self.I_composite_b @ np.array([0.0, 0.0, w_des]))[2]
            self.X_nom[8::9] = yaw_momentum
            X_ter[8] = yaw_momentum
        bounds = np.tile([-self.bx, -self.by, 0, self.bx, self.by, self.bz], (self.horizon, 1))
        assert np.array_equal(bounds[0], BOUNDS_TILE)
        self.mp.create_bound_constraints(bounds, self.fx_max, self.fy_max, self.fz_max)
        self.mp.create_cost_X(np.tile(p.W_X, self.horizon), np.asarray(p.W_X_ter, float), X_ter, self.X_nom)
        self.mp.create_cost_F(np.tile(p.W_F, self.horizon))
        self.X_ter = X_ter

    def optimize(self, q, v, t, v_des, w_des, X_wm=None, F_wm=None, P_wm=None, noise_std=None,
                 mcts_x_y_cnt_loc=None, v_feet_des=None, ee_pos=None, z_height=None, dyn_iters=100):
        """:629-698.  dyn_iters is the ADMM iteration cap the reference hard-codes to 100 (:658)."""
        if v_feet_des is not None:
            raise NotImplementedError("create_cnt_plan_bis is outside the solve path")
        q[0:2] = 0                                               # :633 (mutates the caller's q, as there)
        ori_des = q[3:7] if w_des != 0 else [0, 0, 0, 1]
        R = fk_np._quat_R(np.asarray(q, float)[None, 3:7])[0]
        v_des = R @ np.asarray(v_des, float)
        self.create_cnt_plan(q, v, t, v_des, w_des, noise_std, mcts_x_y_cnt_loc, ee_pos, z_height)
        self.create_costs(q, v, v_des, w_des, ori_des)
        q = np.asarray(q, float).copy()
        q[3:7] /= np.linalg.norm(q[3:7])                         # pin.normalize
        self.kd.optimize(q, v, dyn_iters, 1)
        com_opt, mom_opt = self.mp.return_opt_com(), self.mp.return_opt_mom()
        F_opt = self.mp.return_opt_f().reshape(self.horizon, 3 * len(self.eff_names))
        xs, us = np.array(self.ik.get_xs()), np.array(self.ik.get_us())
        self.f_int = interpolate_plan(F_opt, self.dt_arr, self.size)
        self.xs_int = interpolate_plan(xs, self.dt_arr, self.size)
        self.us_int = interpolate_plan(us, self.dt_arr, self.size)
        self.com_int = interpolate_plan(com_opt, self.dt_arr, self.size)
        self.mom_int = interpolate_plan(mom_opt, self.dt_arr, self.size)
        self.q_traj.append(q)
        self.v_traj.append(v)
        self.xs_traj.append(xs)
        return self.xs_int, self.us_int, self.f_int

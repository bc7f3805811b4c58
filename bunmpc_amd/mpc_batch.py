"""BatchedMpc: SoloMpcGaitGen.optimize (ISL/examples/mpc/abstract_cyclic_gen.py:629-698) for B robots at once, device
resident from the raw states to the 1 kHz plans:

    states x = [q, v], times, desired body-frame velocities
      -> bmpc_wb_plan_batch_device   (FK, contact plan, cost references, IK task blocks)
      -> bmpc_kinodyn_solve_batch_device   (centroidal ADMM + whole-body IK-DDP)
      -> bmpc_interp_batch_device    (xs_int, us_int, f_int)

Everything in between stays in HBM; torch holds the memory and the stream.  w_des = 0 (the data path of the reference's
data generation)."""
import numpy as np

from . import fk_np, problems
from .inverse_kinematics_cpp import as_device_model
from .kinodyn_batch import KinoDynDeviceBatch
from .plan_batch import DeviceWbPlan, interpolate_on_device


class BatchedMpc:
    def __init__(self, model, gait=problems.TROT, ik=problems.TROT_IK, wb=problems.SOLO12_WB, planning_time=0.05, ik_hor_ratio=0.5,
                 dyn_iters=10, device="cuda"):
        self.model, self.gait, self.ik, self.wb = model, gait, ik, wb
        self.dm = as_device_model(model)
        self.device = device
        self.dyn_iters = dyn_iters
        self.H = gait.horizon
        self.T = int(np.round(ik_hor_ratio * gait.gait_horizon * gait.gait_period / gait.gait_dt, 2))     # :128
        self.size = min(self.T, int(planning_time / gait.gait_dt) + 2)                                      # :150-152
        if planning_time > gait.gait_dt:
            self.size -= 1
        k0 = fk_np.kinematics(model, wb.q0[None])
        offs = np.round(fk_np.frame_positions(model, k0, wb.hips)[0] - k0["com"][0], 3)                     # :56-59
        offs[:, 1] += np.array([0.04, -0.04, 0.04, -0.04])                                                 # :61-72
        self.offsets_xy = offs[:, :2]
        self._kb = self._plan = None

    def optimize(self, x, t0, v_des_body, same_rollouts=False):
        """same_rollouts=True: this call replans the rollouts of the previous call with the same batch size (the reference keeps
        one KinoDynMP per rollout, whose FISTA step constants persist from replan to replan: fista.hpp:52); False: B fresh
        rollouts (the data-generation passes).
        x (B,37), t0 (B,), v_des_body (B,3) numpy arrays or device tensors -> dict of device tensors: xs_int (B,R,37), us_int (B,R,18),
        f_int (B,R,12), rows (B,) valid rows of each, plus the raw solution (xs, us, X, F).  The raw solution and `plan` are
        views of buffers the next call with the same batch size reuses: copy what has to outlive it."""
        import torch
        x = x.clone() if isinstance(x, torch.Tensor) else np.array(x, dtype=np.float64)
        x[:, 0:2] = 0.0                                                                                    # :633
        B = x.shape[0]
        if self._kb is not None and self._kb.wb.dyn.B == B:
            # same batch size as the last call: the plan tensors, the solver's workspace and its descriptors are reused
            plan, kb = self._plan, self._kb
            plan.update(x, t0, v_des_body).build()
            kb.carry_step_constants(same_rollouts)
        else:
            plan = DeviceWbPlan(self.dm, self.gait, self.offsets_xy, self.wb.feet, self.ik, x, t0, v_des_body, self.H, self.T,
                                device=self.device).build()
            kb = KinoDynDeviceBatch(self._weights_only_batch(B), self.model, device=self.device, num_iters=self.dyn_iters, plan=plan)
        kb.solve()
        T, H = self.T, self.H
        o = kb.off
        xs = kb.ws[:, o["xs"]:o["xs"] + (T + 1) * 37].reshape(B, T + 1, 37)
        us = kb.ws[:, o["us"]:o["us"] + T * 18].reshape(B, T, 18)
        F = kb.dyn.F.reshape(B, H, 12)
        xs_int, rows = interpolate_on_device(xs, plan.dt, self.size)
        us_int, _ = interpolate_on_device(us, plan.dt, self.size)
        f_int, _ = interpolate_on_device(F, plan.dt, self.size)
        self._kb, self._plan = kb, plan
        return dict(xs_int=xs_int, us_int=us_int, f_int=f_int, rows=rows, xs=xs, us=us, X=kb.dyn.X, F=kb.dyn.F, plan=plan)

    def _weights_only_batch(self, B):
        """the small host-provided part of a WholeBodyBatch: weights, bounds, regularisation reference (the per-problem
        arrays come from the device plan: placeholders of the right type here, never uploaded)"""
        g, ik, H = self.gait, self.ik, self.H
        z = lambda *shape: np.broadcast_to(0.0, shape)       # noqa: E731 -- shape-only stand-ins, no memory behind them
        dyn = problems.Batch(self.wb.name + "_mpc", B, H, 4, self.model.total_mass, g.rho, z(B, H, 4, 4), z(B, H), z(B, 9),
                             z(B, 9 * H), z(B, 9), np.tile(g.W_X, H)[None], g.W_X_ter[None].copy(), np.tile(g.W_F, H)[None],
                             np.tile(problems.BOUNDS_TILE, (H, 1))[None], None, None, self.wb.mu, {})
        x_reg = np.concatenate([np.tile(self.wb.q0, (B, 1)), np.zeros((B, 18))], axis=1)
        return problems.WholeBodyBatch(dyn, z(B, 37), self.T, z(B, self.T + 1, 33), ik["state_wt"][None].copy(), ik["ctrl_wt"][None].copy(),
                                       x_reg, ik["cent_wt"][0], ik["cent_wt"][1], ())

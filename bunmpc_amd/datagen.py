"""One device-resident pass of the reference's data generation, from nominal states to training rows:

    nominal (q, v), t0, desired velocity
      -> contact flags of the current knot                      (bmpc_wb_plan_batch_device, plan of the nominal state)
      -> contact-conditioned perturbation + foot-height rejection    (bmpc_perturb_batch_device; data_collection.py:188-262)
      -> plan inputs of the perturbed state, KinoDynMP.optimize, 1 kHz plan   (BatchedMpc; abstract_cyclic_gen.py:629-698)
      -> inverse-dynamics controller, pd_target action, policy state     (bmpc_id_batch_device; simulation.py:484-528)

What the reference has between the last two steps -- a PyBullet rollout that feeds the measured state back every
millisecond -- is out of scope (simulator), so the rows produced here are the expert's labels ON its own plan:
row r of problem b is (state, action) of the plan's r-th millisecond with the measured state equal to the planned one,
row 0 being the perturbed state itself.  The contact-conditioned goals (`cc_goals`) need the simulator's contact
schedule and are left to the caller."""
import numpy as np

from . import dataset, problems
from .mpc_batch import BatchedMpc
from .perturbation import PerturbationSampler
from .robot_id_controller import id_batch_device


class PlanLabelGenerator:
    def __init__(self, model, gait=problems.TROT, ik=problems.TROT_IK, wb=problems.SOLO12_WB, kp=3.0, kd=0.05,
                 mu=(0.0, 0.0, 0.0, 0.0), sigma=(0.05, 0.1, 0.2, 0.2), planning_time=0.05, dyn_iters=10, draws_per_call=4,
                 device="cuda:0"):
        self.mpc = BatchedMpc(model, gait, ik, wb, planning_time=planning_time, dyn_iters=dyn_iters, device=device)
        self.sampler = PerturbationSampler(self.mpc.dm, wb.feet, mu, sigma, draws_per_call=draws_per_call, device=device)
        self.foot_frames = [model.frame_id(n) for n in wb.feet]
        self.kp, self.kd, self.gait, self.wb = kp, kd, gait, wb
        self.rows_per_plan = int(round(planning_time / 0.001))
        self._nominal = None

    def step(self, q_nom, v_nom, t0, v_des_body, generator=None, perturb=True):
        """q_nom (B,19), v_nom (B,18), t0 (B,), v_des_body (B,3): device tensors.  Returns device tensors
        states (B,R,43), actions (B,R,12), vc_goals (B,R,5), and q0 / v0 (the perturbed initial states), `rejected`
        (indices whose draws were all refused: those keep their nominal state)."""
        import torch
        from .plan_batch import DeviceWbPlan
        m = self.mpc
        B = q_nom.shape[0]
        rejected = torch.empty(0, dtype=torch.long, device=q_nom.device)
        if perturb:
            x = torch.cat([q_nom, v_nom], dim=1)
            x[:, 0:2] = 0.0
            if self._nominal is not None and self._nominal.B == B:
                nominal = self._nominal.update(x, t0, v_des_body).build()
            else:
                nominal = self._nominal = DeviceWbPlan(m.dm, m.gait, m.offsets_xy, m.wb.feet, m.ik, x, t0, v_des_body, m.H, m.T,
                                                       device=m.device).build()
            q0, v0, rejected = self.sampler.sample(q_nom, v_nom, nominal.cnt_plan[:, 0, :, 0], generator=generator)
        else:
            q0, v0 = q_nom, v_nom
        sol = m.optimize(torch.cat([q0, v0], dim=1), t0, v_des_body)
        R = min(self.rows_per_plan, int(sol["rows"].min()))
        xs = sol["xs_int"][:, :R].reshape(B * R, 37)
        us = sol["us_int"][:, :R].reshape(B * R, 18)
        f = sol["f_int"][:, :R].reshape(B * R, 12)
        out = id_batch_device(m.dm, self.foot_frames, self.kp, self.kd, xs[:, :19], xs[:, 19:], us, f, want=("action", "state"))
        t = (t0[:, None] + 0.001 * torch.arange(R, device=t0.device, dtype=torch.float64)[None, :])
        vc = torch.zeros((B, R, dataset.VC_GOAL_WIDTH), dtype=torch.float64, device=t0.device)
        vc[:, :, 0] = torch.remainder(t, self.gait.gait_period) / self.gait.gait_period
        vc[:, :, 1:3] = v_des_body[:, None, 0:2]
        vc[:, :, 4] = dataset.GAIT_VALUE.get(self.gait.name, 0.0)
        return dict(states=out["state"].reshape(B, R, 43), actions=out["action"].reshape(B, R, 12), vc_goals=vc, q0=q0, v0=v0,
                    rejected=rejected, solution=sol)

"""InverseDynamicsController (`ISL/examples/controllers/robot_id_controller.py:12-86`) on the GPU library, plus the batched
output stage the data path needs: torques, PD-target actions and the 43-entry policy state rows
(`ISL/examples/iterative_algorithm/simulation.py:156-175, 484-528`) for many samples in one launch
(`bmpc_id_batch_device`, csrc/id_ctrl.hip).  Same method names and argument meaning as the reference class; `robot` is a
RobotModel / URDF path / DeviceModel where the reference takes a pinocchio wrapper.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from .inverse_kinematics_cpp import as_device_model

N_JOINTS, STATE_WIDTH = 12, 43


def _gain(g):
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    if g.size == 1:
        g = np.repeat(g, N_JOINTS)
    if g.size != N_JOINTS:
        raise ValueError("gains: one value or one per actuated joint")
    return g


def id_batch_device(dev_model, foot_frames, kp, kd, q_des, v_des, a_des, f, q=None, v=None, want=("tau_ff", "tau_fb", "action", "state")):
    """Rows on the GPU (torch float64 tensors; any row stride, unit column stride): q_des (n,19), v_des (n,18), a_des (n,18),
    f (n,12), optionally the measured q (n,19), v (n,18).  Returns a dict of dense tensors for the names in `want`."""
    import torch
    n = q_des.shape[0]
    d = _lib.IdBatch()
    d.n, d.model = n, dev_model.h
    d.foot_frame[:] = [int(x) for x in foot_frames]
    d.kp[:] = list(_gain(kp))
    d.kd[:] = list(_gain(kd))
    rows = dict(q_des=(q_des, 19), v_des=(v_des, 18), a_des=(a_des, 18), f=(f, 12))
    if (q is None) != (v is None):
        raise ValueError("q and v go together")
    if q is not None:
        rows.update(q=(q, 19), v=(v, 18))
    for name, (t, w) in rows.items():
        if t.dtype != torch.float64 or not t.is_cuda or t.dim() != 2 or t.shape != (n, w) or (n > 0 and t.stride(1) != 1):
            raise ValueError("%s: expected a CUDA float64 tensor of shape (%d, %d) with contiguous rows" % (name, n, w))
        setattr(d, name, t.data_ptr())
        setattr(d, "s_" + name, t.stride(0) if n > 1 else w)
    out = {}
    for name in want:
        out[name] = torch.empty((n, STATE_WIDTH if name == "state" else N_JOINTS), dtype=torch.float64, device=q_des.device)
        setattr(d, name, out[name].data_ptr())
    stream = torch.cuda.current_stream(q_des.device).cuda_stream
    _lib.check(_lib.lib().bmpc_id_batch_device(C.byref(d), C.c_void_p(stream)))
    return out


class InverseDynamicsController:
    def __init__(self, robot, eff_arr, pinModel=None, pinData=None, real_robot=False, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("bunmpc_amd needs a GPU: there is no CPU fallback")
        self.torch, self.device = torch, torch.device(device)
        self.dev_model = as_device_model(robot if pinModel is None else pinModel)
        self.model = self.dev_model.model
        self.nq, self.nv = self.model.nq, self.model.nv
        self.robot_mass = self.model.total_mass
        self.eff_arr = list(eff_arr)
        self.foot_frames = [self.model.frame_id(n) for n in self.eff_arr]

    def set_gains(self, kp, kd):
        self.kp, self.kd = kp, kd

    def _dev(self, a, w):
        return self.torch.as_tensor(np.asarray(a, dtype=np.float64).reshape(-1, w), device=self.device)

    def compute_id_torques(self, q, v, a):
        """rnea(q, v, a): the actuated rows are computed on the GPU; the six base rows are not part of the controller's
        output (`id_joint_torques` drops them, :81) and are returned as NaN"""
        r = id_batch_device(self.dev_model, self.foot_frames, 1.0, 0.0, self._dev(q, 19), self._dev(v, 18), self._dev(a, 18),
                            self.torch.zeros((1, 12), dtype=self.torch.float64, device=self.device), want=("tau_ff",))
        return np.concatenate([np.full(6, np.nan), r["tau_ff"][0].cpu().numpy()])

    def id_joint_torques(self, q, dq, des_q, des_v, des_a, fff):
        assert len(q) == self.nq
        r = id_batch_device(self.dev_model, self.foot_frames, self.kp, self.kd, self._dev(des_q, 19), self._dev(des_v, 18),
                            self._dev(des_a, 18), self._dev(fff, 12), self._dev(q, 19), self._dev(dq, 18), want=("tau_ff", "tau_fb"))
        return r["tau_ff"][0].cpu().numpy(), r["tau_fb"][0].cpu().numpy()

    def rows(self, xs_int, us_int, f_int, q=None, v=None, want=("tau_ff", "tau_fb", "action", "state")):
        """the batched form: xs_int (n,37) / us_int (n,18) / f_int (n,12) rows of 1 kHz plans on the GPU (views are fine)"""
        return id_batch_device(self.dev_model, self.foot_frames, self.kp, self.kd, xs_int[:, :19], xs_int[:, 19:], us_int, f_int,
                               q, v, want)

"""Harness inputs built on the GPU (bmpc_plan_batch_device in include/bunmpc.h): contact plan, swing flags, dt, X_nom,
X_ter for B problems from their current CoM / feet / time / desired velocities, left in HBM for
`bmpc_biconvex_solve_batch_device`.  Device counterpart of problems.contact_plan / centroidal_costs."""
import ctypes as C

import numpy as np

from . import _lib


def gait_struct(gait, offsets_xy):
    g = _lib.GaitParams()
    g.gait_period, g.gait_dt, g.gait_horizon, g.nom_ht = gait.gait_period, gait.gait_dt, gait.gait_horizon, gait.nom_ht
    for j in range(4):
        g.stance_percent[j], g.phase_offset[j] = gait.stance_percent[j], gait.phase_offset[j]
        g.offsets_xy[j][0], g.offsets_xy[j][1] = offsets_xy[j][0], offsets_xy[j][1]
    for k in range(3):
        g.ori_correction[k] = gait.ori_correction[k]
    return g


class DevicePlan:
    """Builds (and keeps) the plan tensors of one batch on `device`.  Inputs are numpy arrays or torch tensors:
    t0 (B,), com (B,3), feet0 (B,4,3), v_des (B,3), w_des (B,), x_init (B,9), optional amom (B,3), hip_off (B,4,2),
    gait_id (B,) into `gaits` (list of GaitParams with the robot's hip offsets)."""

    def __init__(self, gaits, offsets_xy, H, t0, com, feet0, v_des, w_des, x_init, amom=None, hip_off=None, gait_id=None,
                 device="cuda"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePlan needs a GPU")
        self.torch, self.device = torch, torch.device(device)
        f64 = torch.float64

        def up(a, dtype=f64):
            return None if a is None else torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a, dtype=dtype, device=self.device).contiguous()

        B = int(np.shape(t0)[0])
        self.B, self.H = B, H
        raw = (_lib.GaitParams * len(gaits))(*[gait_struct(g, offsets_xy) for g in gaits])
        self.gaits = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).to(self.device)
        self.inp = dict(t0=up(t0), com=up(com), feet0=up(feet0), v_des=up(v_des), w_des=up(w_des), x_init=up(x_init),
                        amom=up(amom), hip_off=up(hip_off), gait_id=up(gait_id, torch.int32))
        self.cnt_plan = torch.empty((B, H, 4, 4), dtype=f64, device=self.device)
        self.swing_time = torch.empty((B, H, 4), dtype=f64, device=self.device)
        self.dt = torch.empty((B, H), dtype=f64, device=self.device)
        self.X_nom = torch.empty((B, 9 * H), dtype=f64, device=self.device)
        self.X_ter = torch.empty((B, 9), dtype=f64, device=self.device)
        d = _lib.PlanBatch()
        d.B, d.n_col, d.n_gaits = B, H, len(gaits)
        d.gaits = self.gaits.data_ptr()
        for k, v in self.inp.items():
            setattr(d, k, None if v is None else v.data_ptr())
        for k in ("cnt_plan", "swing_time", "dt", "X_nom", "X_ter"):
            setattr(d, k, getattr(self, k).data_ptr())
        self.desc = d

    def build(self):
        """asynchronous on torch's current stream"""
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().bmpc_plan_batch_device(C.byref(self.desc), C.c_void_p(stream)))
        return self


class DeviceWbPlan:
    """Whole-body front end on the GPU (bmpc_wb_plan_batch_device): from states x = [q, v], times and body-frame
    desired velocities to the centroidal batch inputs and the IK task blocks, all left in HBM.  Device counterpart of
    problems.make_wb_batch (weights / regularisation references stay small host-provided arrays)."""

    def __init__(self, dev_model, gait, offsets_xy, feet, ik, x, t0, v_des_body, H, T, device="cuda"):
        import torch
        self.torch, self.device = torch, torch.device(device)
        f64 = torch.float64

        def up(a):      # numpy arrays are uploaded, device tensors are taken as they are
            if isinstance(a, torch.Tensor):
                return a.to(device=self.device, dtype=f64).contiguous()
            return torch.as_tensor(np.ascontiguousarray(a), dtype=f64, device=self.device).contiguous()

        B = int(x.shape[0])
        self.B, self.H, self.T = B, H, T
        raw = (_lib.GaitParams * 1)(gait_struct(gait, offsets_xy))
        self.gait = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).to(self.device)
        self.x, self.t0, self.v_des_body = up(x), up(t0), up(v_des_body)

        def z(*shape):
            return torch.empty(shape, dtype=f64, device=self.device)

        self.com, self.feet0, self.v_des, self.w_des = z(B, 3), z(B, 4, 3), z(B, 3), z(B)
        self.hip_off, self.amom, self.x_init = z(B, 4, 2), z(B, 3), z(B, 9)
        self.cnt_plan, self.swing_time, self.dt = z(B, H, 4, 4), z(B, H, 4), z(B, H)
        self.X_nom, self.X_ter, self.ik_tasks = z(B, 9 * H), z(B, 9), z(B, T + 1, 33)
        self.inp = dict(x_init=self.x_init)      # DeviceBatch(plan=...) reads x_init from here
        d = _lib.WbPlanBatch()
        d.B, d.n_col, d.ik_col = B, H, T
        d.model, d.gait = dev_model.h, self.gait.data_ptr()
        for j in range(4):
            d.foot_frame[j] = dev_model.model.frame_id(feet[j])
        d.step_ht = gait.step_ht
        for k in range(2):
            d.swing_wt[k], d.cent_wt[k], d.reg_wt[k] = ik["swing_wt"][k], ik["cent_wt"][k], ik["reg_wt"][k]
        for k in ("x", "t0", "v_des_body", "com", "feet0", "v_des", "w_des", "hip_off", "amom", "x_init", "cnt_plan", "swing_time",
                  "dt", "X_nom", "X_ter", "ik_tasks"):
            setattr(d, k, getattr(self, k).data_ptr())
        self.desc = d
        self.dev_model = dev_model

    def update(self, x, t0, v_des_body):
        """new states / times / desired velocities for the same batch size, copied into the tensors the descriptor points
        at (nothing is allocated: the steady state of a generator that plans batch after batch)"""
        torch = self.torch
        for dst, src in ((self.x, x), (self.t0, t0), (self.v_des_body, v_des_body)):
            dst.copy_(src if isinstance(src, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(src), dtype=torch.float64))
        return self

    def build(self):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().bmpc_wb_plan_batch_device(C.byref(self.desc), C.c_void_p(stream)))
        return self


def interpolate_on_device(knots, dt, size, step=0.001):
    """1 kHz plan of a batch of knot trajectories (torch tensors on the GPU): knots (B, n, w), dt (B, H) ->
    (out (B, max_rows, w), rows (B,)); out[b, :rows[b]] equals cyclic_gen.interpolate_plan(knots[b], dt[b], size)."""
    import torch
    B, n, w = knots.shape
    knots, dt = knots.contiguous(), dt.contiguous()
    max_rows = int(size * (int(float(dt.max()) / step) + 1))
    out = torch.zeros((B, max_rows, w), dtype=torch.float64, device=knots.device)
    rows = torch.zeros(B, dtype=torch.int32, device=knots.device)
    d = _lib.InterpBatch()
    d.B, d.n_knots, d.width, d.size, d.max_rows, d.dt_stride, d.step = B, n, w, size, max_rows, dt.shape[1], step
    d.knots, d.dt, d.out, d.rows = knots.data_ptr(), dt.data_ptr(), out.data_ptr(), rows.data_ptr()
    stream = torch.cuda.current_stream(knots.device).cuda_stream
    _lib.check(_lib.lib().bmpc_interp_batch_device(C.byref(d), C.c_void_p(stream)))
    return out, rows

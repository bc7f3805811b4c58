"""Batched KinoDynMP.optimize on one MI355X: centroidal state of (q, v) -> centroidal ADMM ->
tracking references -> whole-body IK-DDP, for B independent problems, everything resident in HBM
(bmpc_kinodyn_solve_batch_device in include/bunmpc.h).  torch provides device memory and the stream."""
import ctypes as C

import numpy as np

from . import _lib
from .batch import DeviceBatch
from .inverse_kinematics_cpp import as_device_model


class KinoDynDeviceBatch:
    def __init__(self, wb, model, device="cuda", num_iters=10, maxit=150, ddp_maxiter=100, plan=None, use_active_list=True, schedule=None,
                 keep_hist=False):
        """plan: a plan_batch.DeviceWbPlan whose tensors replace the host-built centroidal inputs and IK task blocks of
        `wb` (weights and regularisation references still come from wb).  schedule: dict of bmpc_ik_sched_t fields for this
        batch's DDP loops (0 = process default, < 0 = never), e.g. {"gains_wave_below": -1}; no effect on results"""
        import torch
        self.torch = torch
        self.wb = wb
        self.dm = as_device_model(model)
        self.dyn = DeviceBatch(wb.dyn, device=device, num_iters=num_iters, maxit=maxit, plan=plan, keep_hist=keep_hist)
        self.device = self.dyn.device
        B, T = wb.dyn.B, wb.ik_T
        f64 = torch.float64

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

        self.x = up(wb.x) if plan is None else plan.x
        self.tasks = up(wb.ik_tasks) if plan is None else plan.ik_tasks
        self.state_w, self.ctrl_w, self.x_reg = up(wb.state_w), up(wb.ctrl_w), up(wb.x_reg)
        self.dt_ik = up(wb.dyn.dt[:, :T]) if plan is None else plan.dt[:, :T].contiguous()
        lib = _lib.lib()
        self.ws_doubles = lib.bmpc_ik_workspace_doubles(T)
        off = (C.c_long * 8)()
        lib.bmpc_ik_layout(T, off)
        self.off = dict(zip(("xs", "us", "scal", "K", "k", "fs", "Lx", "Lqq"), list(off)))
        t_off, t_it, t_w = C.c_long(0), C.c_int(0), C.c_int(0)
        lib.bmpc_ik_layout_trace(T, C.byref(t_off), C.byref(t_it), C.byref(t_w))
        self.trace_off, self.trace_iters, self.trace_width = t_off.value, t_it.value, t_w.value
        self.ws = torch.zeros((B, self.ws_doubles), dtype=f64, device=self.device)
        self.active = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.active_list = torch.zeros(lib.bmpc_ik_active_list_ints(B), dtype=torch.int32, device=self.device)
        self.iters_run = C.c_int(0)
        d = _lib.KinoDynBatch()
        C.memmove(C.byref(d.dyn), C.byref(self.dyn.desc), C.sizeof(_lib.Batch))
        ik = d.ik
        ik.B, ik.n_col, ik.maxiter, ik.model = B, T, ddp_maxiter, self.dm.h
        ik.x0, ik.dt, ik.tasks = self.x.data_ptr(), self.dt_ik.data_ptr(), self.tasks.data_ptr()
        ik.state_w, ik.x_reg, ik.ctrl_w = self.state_w.data_ptr(), self.x_reg.data_ptr(), self.ctrl_w.data_ptr()
        ik.s_state_w = 0 if wb.state_w.shape[0] == 1 else 36
        ik.s_ctrl_w = 0 if wb.ctrl_w.shape[0] == 1 else 18
        ik.ws, ik.active = self.ws.data_ptr(), self.active.data_ptr()
        ik.active_list = self.active_list.data_ptr() if use_active_list else None
        ik.iters_run = C.addressof(self.iters_run)
        d.x = self.x.data_ptr()
        self.desc = d
        self.set_schedule(**(schedule or {}))

    def set_schedule(self, **fields):
        """per-batch scheduling thresholds of the DDP loop (bmpc_ik_sched_t)"""
        for k, v in fields.items():
            if k not in ("spec_below", "all_steps_below", "gains_wave_below", "express_cap", "debug_inject"):
                raise KeyError(k)
            setattr(self.desc.ik.sched, k, int(v))

    def carry_step_constants(self, on=True):
        """True: the next solves are further optimize calls of the SAME KinoDynMP objects (successive replans of the same
        rollouts): iterates reset, FISTA's step constants carried (bmpc_batch_t.cold_start = 2); False: fresh objects"""
        self.desc.dyn.cold_start = 2 if on else 1

    def solve(self):
        """one full batch of KinoDynMP.optimize; synchronises once per DDP iteration (active counter)"""
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        if self.dyn.hist is not None:      # rows of ADMM iterations that do not run keep their NaN / -1
            self.dyn.hist.fill_(float("nan"))
            self.dyn.trace.fill_(-1)
        _lib.check(_lib.lib().bmpc_kinodyn_solve_batch_device(C.byref(self.desc), C.c_void_p(stream)))

    def solve_ik_only(self):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().bmpc_ik_solve_batch_device(C.byref(self.desc.ik), C.c_void_p(stream)))

    def results(self):
        self.torch.cuda.synchronize(self.device)
        out = self.dyn.results()
        T = self.wb.ik_T
        ws = self.ws.cpu().numpy()
        o = self.off
        out["xs"] = ws[:, o["xs"]:o["xs"] + (T + 1) * 37].reshape(-1, T + 1, 37)
        out["us"] = ws[:, o["us"]:o["us"] + T * 18].reshape(-1, T, 18)
        sc = ws[:, o["scal"]:o["scal"] + 16]
        out["ik_cost"], out["ik_stop"] = sc[:, 0], sc[:, 4]
        out["ik_iters"], out["ik_status"] = sc[:, 8].astype(np.int64), sc[:, 10].astype(np.int64)
        out["ik_wide_line_search"] = sc[:, 12] != 0     # flagged: ran its later line searches with all ten step lengths at once
        out["ddp_loop_iters"] = self.iters_run.value
        # DDP iterations a problem ran inside the persistent fused kernel (the express lane); 0 = solved by the batch's own kernels
        oq = o["k"] + T * 18
        out["ik_fused_iters"] = ws[:, oq + 3].astype(np.int64) if T * 18 >= 8 else np.zeros(ws.shape[0], dtype=np.int64)
        # rows [iteration][cost, regularisation, accepted step length (0 = none), |Q_u|^2]; rows past ik_iters are stale
        out["ik_trace"] = ws[:, self.trace_off:self.trace_off + self.trace_iters * self.trace_width].reshape(-1, self.trace_iters, self.trace_width)
        return out

"""Drop-in for the reference's pybind module `inverse_kinematics_cpp`
(iterative_supervised_learning/srcpy/ik/inverse_kinematics.cpp:16-41): `InverseKinematics` with the
same method names and argument meaning, backed by the C-ABI (include/bunmpc.h); `optimize` runs the
batched DDP kernels (bunmpc_amd/csrc/ik_ddp.hip) with B = 1.

`fid` arguments are indices into the model's frame list; bunmpc_amd.urdf_model.load_urdf numbers
frames in the order pinocchio's URDF parser creates them, so `pin_model.getFrameId(name)` values
carry over (a frame *name* is accepted as well)."""
import ctypes as C
import os

import numpy as np

from . import _lib, urdf_model


class DeviceModel:
    """bmpc_model_t built from a urdf_model.RobotModel (kept alive by whoever uses the handle)."""

    def __init__(self, model):
        self.model = model
        lib = _lib.lib()
        fr = list(model.frames.values())
        fbody = np.array([f[0] for f in fr], dtype=np.int32)
        fp = np.ascontiguousarray([f[2] for f in fr], dtype=np.float64)
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in
                (model.R.reshape(model.nj, 9), model.p, model.axis, model.mass, model.com, model.inertia.reshape(-1, 9))]
        parent = np.ascontiguousarray(model.parent, dtype=np.int32)
        self._destroy = lib.bmpc_model_destroy
        self.h = lib.bmpc_model_create(model.nj, parent.ctypes.data, *[a.ctypes.data for a in arrs], len(fr),
                                       fbody.ctypes.data, fp.ctypes.data)
        if not self.h:
            raise _lib.BmpcError(_lib.BAD_ARG, _lib.last_error())

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self._destroy(h)       # bound at construction: module globals may be gone at interpreter exit


def as_device_model(m):
    """accepts a URDF path, a RobotModel or a DeviceModel"""
    if isinstance(m, DeviceModel):
        return m
    if isinstance(m, (str, os.PathLike)):
        p = str(m)
        m = urdf_model.RobotModel.from_json(open(p).read()) if p.endswith(".json") else urdf_model.load_urdf(p)
    return DeviceModel(m)


def _f64(a, n=None, name="array"):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if n is not None and a.size != n:
        raise ValueError("%s: expected %d values, got %d" % (name, n, a.size))
    return a


class InverseKinematics:
    """ik::InverseKinematics (ISL/include/ik/inverse_kinematics.hpp:43-133)."""

    def __init__(self, rmodel_path, n_col, _handle=None, _owner=None, _dmodel=None):
        self._lib = _lib.lib()
        self._owner = _owner
        self._dm = _dmodel if _dmodel is not None else as_device_model(rmodel_path)
        self._owned = _handle is None
        self._h = self._lib.bmpc_ik_create(self._dm.h, int(n_col)) if _handle is None else _handle
        if not self._h:
            raise _lib.BmpcError(_lib.BAD_ARG, _lib.last_error())
        self.n_col = int(n_col)
        self.nq, self.nv = self._dm.model.nq, self._dm.model.nv

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self, "_owned", False):
            self._lib.bmpc_ik_destroy(h)

    def _fid(self, fid):
        return self._dm.model.frame_id(fid) if isinstance(fid, str) else int(fid)

    def setup_costs(self, dt):
        dt = _f64(dt)
        _lib.check(self._lib.bmpc_ik_setup_costs(self._h, dt.ctypes.data, dt.size))

    def optimize(self, x0):
        x0 = _f64(x0, self.nq + self.nv, "x0")
        _lib.check(self._lib.bmpc_ik_optimize(self._h, x0.ctypes.data))

    def get_xs(self):
        out = np.zeros((self.n_col + 1, self.nq + self.nv))
        _lib.check(self._lib.bmpc_ik_get_xs(self._h, out.ctypes.data))
        return [r.copy() for r in out]

    def get_us(self):
        out = np.zeros((self.n_col, self.nv))
        _lib.check(self._lib.bmpc_ik_get_us(self._h, out.ctypes.data))
        return [r.copy() for r in out]

    def return_opt_com(self):
        out = np.zeros((self.n_col + 1, 3))
        _lib.check(self._lib.bmpc_ik_return_opt_com(self._h, out.ctypes.data))
        return out

    def return_opt_mom(self):
        out = np.zeros((self.n_col + 1, 6))
        _lib.check(self._lib.bmpc_ik_return_opt_mom(self._h, out.ctypes.data))
        return out

    def add_position_tracking_task(self, fid, sn, en, traj, wt, cost_name):
        t = _f64(traj, 3, "traj")
        _lib.check(self._lib.bmpc_ik_add_position_tracking_task(self._h, self._fid(fid), int(sn), int(en), t.ctypes.data,
                                                                float(wt), cost_name.encode()))

    def add_position_tracking_task_single(self, fid, traj, wt, cost_name, time_step):
        t = _f64(traj, 3, "traj")
        _lib.check(self._lib.bmpc_ik_add_position_tracking_task_single(self._h, self._fid(fid), t.ctypes.data, float(wt),
                                                                       cost_name.encode(), int(time_step)))

    def add_terminal_position_tracking_task(self, fid, traj, wt, cost_name):
        t = _f64(traj, 3, "traj")
        _lib.check(self._lib.bmpc_ik_add_terminal_position_tracking_task(self._h, self._fid(fid), t.ctypes.data, float(wt),
                                                                         cost_name.encode()))

    def add_velocity_tracking_task(self, fid, sn, en, traj, wt, cost_name):
        _lib.check(self._lib.bmpc_ik_add_velocity_tracking_task(self._h))

    def add_com_position_tracking_task(self, sn, en, traj, wt, cost_name, isTerminal=False):
        t = np.atleast_2d(_f64(traj)).reshape(-1, 3)
        _lib.check(self._lib.bmpc_ik_add_com_position_tracking_task(self._h, int(sn), int(en), t.ctypes.data, t.shape[0],
                                                                    float(np.asarray(wt).reshape(-1)[0]), cost_name.encode(), int(bool(isTerminal))))

    def add_centroidal_momentum_tracking_task(self, sn, en, traj, wt, cost_name, isTerminal=False):
        t = np.atleast_2d(_f64(traj)).reshape(-1, 6)
        _lib.check(self._lib.bmpc_ik_add_centroidal_momentum_tracking_task(self._h, int(sn), int(en), t.ctypes.data, t.shape[0],
                                                                           float(np.asarray(wt).reshape(-1)[0]), cost_name.encode(), int(bool(isTerminal))))

    def add_state_regularization_cost(self, sn, en, wt, cost_name, stateWeights, x_reg, isTerminal=False):
        w, x = _f64(stateWeights, 2 * self.nv, "stateWeights"), _f64(x_reg, self.nq + self.nv, "x_reg")
        _lib.check(self._lib.bmpc_ik_add_state_regularization_cost(self._h, int(sn), int(en), float(wt), cost_name.encode(),
                                                                   w.ctypes.data, x.ctypes.data, int(bool(isTerminal))))

    def add_state_regularization_cost_single(self, time_step, wt, cost_name, stateWeights, x_reg):
        w, x = _f64(stateWeights, 2 * self.nv, "stateWeights"), _f64(x_reg, self.nq + self.nv, "x_reg")
        _lib.check(self._lib.bmpc_ik_add_state_regularization_cost_single(self._h, int(time_step), float(wt), cost_name.encode(),
                                                                          w.ctypes.data, x.ctypes.data))

    def add_ctrl_regularization_cost(self, sn, en, wt, cost_name, controlWeights, u_reg, isTerminal):
        w, u = _f64(controlWeights, self.nv, "controlWeights"), _f64(u_reg, self.nv, "u_reg")
        _lib.check(self._lib.bmpc_ik_add_ctrl_regularization_cost(self._h, int(sn), int(en), float(wt), cost_name.encode(),
                                                                  w.ctypes.data, u.ctypes.data, int(bool(isTerminal))))

    def add_ctrl_regularization_cost_single(self, time_step, wt, cost_name, controlWeights, u_reg):
        w, u = _f64(controlWeights, self.nv, "controlWeights"), _f64(u_reg, self.nv, "u_reg")
        _lib.check(self._lib.bmpc_ik_add_ctrl_regularization_cost_single(self._h, int(time_step), float(wt), cost_name.encode(),
                                                                         w.ctypes.data, u.ctypes.data))

    # additive
    def last_stats(self):
        it, st, c, s = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        _lib.check(self._lib.bmpc_ik_last_stats(self._h, C.byref(it), C.byref(st), C.byref(c), C.byref(s)))
        return dict(iters=it.value, status=st.value, cost=c.value, stop=s.value)

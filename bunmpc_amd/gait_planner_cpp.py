"""Drop-in for the reference's pybind module `gait_planner_cpp`
(iterative_supervised_learning/srcpy/gait_planner/py_gait_planner.cpp:15-36): class
`GaitPlanner` with the same overloads, backed by the C-ABI (include/bunmpc.h)."""
import ctypes as C

import numpy as np

from . import _lib


class GaitPlanner:
    def __init__(self, gait_period, stance_percent, phase_offset, step_height):
        sp = np.ascontiguousarray(stance_percent, dtype=np.float64).reshape(-1)
        po = np.ascontiguousarray(phase_offset, dtype=np.float64).reshape(-1)
        if sp.shape != po.shape:
            raise TypeError("stance_percent and phase_offset must have the same length")
        self._n = sp.shape[0]
        self._h = _lib.lib().bmpc_gait_create(float(gait_period), sp.ctypes.data, po.ctypes.data,
                                              self._n, float(step_height))
        if not self._h:
            raise _lib.BmpcError(_lib.BAD_ARG, _lib.last_error())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().bmpc_gait_destroy(h)

    # get_phase(t, foot) -> int | get_phase(t) -> (n_eff,) int      (py_gait_planner.cpp:25-26)
    def get_phase(self, time_in, foot_ID=None):
        if foot_ID is None:
            out = np.zeros(self._n, dtype=np.int32)
            _lib.check(_lib.lib().bmpc_gait_get_phase_all(self._h, float(time_in), out.ctypes.data))
            return out
        v = C.c_int()
        _lib.check(_lib.lib().bmpc_gait_get_phase(self._h, float(time_in), int(foot_ID), C.byref(v)))
        return v.value

    # get_phi(t, foot) -> float | get_phi(t) -> (n_eff,)             (:27-28)
    def get_phi(self, time_in, foot_ID=None):
        if foot_ID is None:
            out = np.zeros(self._n)
            _lib.check(_lib.lib().bmpc_gait_get_phi_all(self._h, float(time_in), out.ctypes.data))
            return out
        v = C.c_double()
        _lib.check(_lib.lib().bmpc_gait_get_phi(self._h, float(time_in), int(foot_ID), C.byref(v)))
        return v.value

    # get_percent_in_phase(t, foot) | (t)                            (:29-32)
    def get_percent_in_phase(self, time_in, foot_ID=None):
        if foot_ID is None:
            out = np.zeros(self._n)
            _lib.check(_lib.lib().bmpc_gait_get_percent_in_phase_all(self._h, float(time_in), out.ctypes.data))
            return out
        v = C.c_double()
        _lib.check(_lib.lib().bmpc_gait_get_percent_in_phase(self._h, float(time_in), int(foot_ID), C.byref(v)))
        return v.value

    # get_contact_phase_plan(plan, t, dt) -> int matrix              (:33)
    def get_contact_phase_plan(self, contact_phase_plan, time_in, dt):
        rows = np.asarray(contact_phase_plan).shape[0]
        out = np.zeros((rows, self._n), dtype=np.int32)
        _lib.check(_lib.lib().bmpc_gait_get_contact_phase_plan(self._h, rows, float(time_in), float(dt),
                                                               out.ctypes.data))
        return out

    # the reference binds its two setters under the name get_phi by mistake (:34-35); exposed
    # here under their C++ names
    def set_step_height(self, step_height):
        _lib.check(_lib.lib().bmpc_gait_set_step_height(self._h, float(step_height)))

    def set_stance_percent(self, lf, lh, rf, rh):
        _lib.check(_lib.lib().bmpc_gait_set_stance_percent(self._h, float(lf), float(lh), float(rf), float(rh)))

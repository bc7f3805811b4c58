"""The quaternion versions of the state operators the forward pass uses (rbd_device.h: state_diff_q, state_integrate_q: no
rotation matrices, Newton-refined reciprocals, an own arctangent) against the rotation-matrix versions on the device and
against the numpy restatement (oracle/rbd_np.py), over small and large rotations."""
import os

import numpy as np
import pytest

from bunmpc_amd import _lib, urdf_model
from oracle import rbd_np as rb

pytestmark = pytest.mark.gpu
ROBOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots", "solo12.json")


def test_quaternion_state_operators():
    model = urdf_model.RobotModel.from_json(open(ROBOT).read())
    rng = np.random.default_rng(11)
    n = 256
    scales = np.concatenate([np.full(64, 1e-9), np.full(64, 1e-4), np.full(64, 0.05), np.full(48, 1.0), np.full(16, 3.0)])
    x0 = np.array([np.concatenate([rb.integrate(model, rb.neutral(model), rng.standard_normal(18)), rng.standard_normal(18)]) for _ in range(n)])
    dx = scales[:, None] * rng.standard_normal((n, 36))
    x1 = np.array([rb.state_integrate(model, x0[i], scales[i] * rng.standard_normal(36)) for i in range(n)])
    x1[::7] = x0[::7]                                           # identical states: the zero-rotation branch
    out = [np.zeros((n, 36)), np.zeros((n, 36)), np.zeros((n, 37)), np.zeros((n, 37))]
    f = lambda a: np.ascontiguousarray(a).ctypes.data           # noqa: E731
    _lib.check(_lib.lib().bmpc_ik_selftest_state_ops(f(x0), f(x1), f(dx), n, *[o.ctypes.data for o in out]))
    dq, dr, iq, ir = out
    ref_d = np.array([rb.state_diff(model, x0[i], x1[i]) for i in range(n)])
    ref_i = np.array([rb.state_integrate(model, x0[i], dx[i]) for i in range(n)])
    tol = 1e-13 * np.maximum(1.0, np.abs(ref_d).max(axis=1, keepdims=True))
    assert np.all(np.abs(dq - ref_d) <= tol), np.abs(dq - ref_d).max()
    assert np.all(np.abs(dq - dr) <= tol)
    # q and -q are the same rotation: compare the integrated quaternions up to sign
    for got in (iq, ir):
        sign = np.sign(np.sum(got[:, 3:7] * ref_i[:, 3:7], axis=1, keepdims=True))
        g = got.copy()
        g[:, 3:7] *= sign
        assert np.abs(g - ref_i).max() < 1e-13
    assert np.abs(np.linalg.norm(iq[:, 3:7], axis=1) - 1.0).max() < 1e-15

"""bmpc_id_batch_device (csrc/id_ctrl.hip) against oracle/id_np.py: torques, PD-target actions and policy state rows.
Floating-point path: tolerance 1e-11 relative to the largest torque of the sample (the two differ in operation order)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROBOTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots")
FEET = {"solo12": ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"], "go2": ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]}


def _setup(robot):
    from bunmpc_amd import urdf_model
    return urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, robot + ".json")).read())


def _samples(model, n, seed, unit_quat=False):
    from oracle import rbd_np
    rng = np.random.default_rng(seed)
    qd, q = np.zeros((n, 19)), np.zeros((n, 19))
    for i in range(n):
        base = rbd_np.neutral(model)
        base[:3] = rng.normal(0, 0.5, 3)
        qd[i] = rbd_np.integrate(model, base, rng.normal(0, 0.6, 18))
        q[i] = rbd_np.integrate(model, qd[i], rng.normal(0, 0.05, 18))
        if not unit_quat:
            qd[i, 3:7] *= 1.0 - 0.01 * rng.random()       # what linear interpolation between knots does
    return dict(q_des=qd, v_des=rng.normal(0, 1.0, (n, 18)), a_des=rng.normal(0, 5.0, (n, 18)), f=rng.normal(0, 8.0, (n, 12)),
                q=q, v=rng.normal(0, 1.0, (n, 18)))


def _oracle(model, feet, s, kp, kd, measured=True):
    from oracle import id_np
    ctrl = id_np.InverseDynamicsController(model, feet)
    ctrl.set_gains(kp, kd)
    n = s["q_des"].shape[0]
    out = dict(tau_ff=np.zeros((n, 12)), tau_fb=np.zeros((n, 12)), action=np.zeros((n, 12)), state=np.zeros((n, 43)))
    for i in range(n):
        q, v = (s["q"][i], s["v"][i]) if measured else (s["q_des"][i], s["v_des"][i])
        ff, fb = ctrl.id_joint_torques(q, v, s["q_des"][i], s["v_des"][i], s["a_des"][i], s["f"][i])
        out["tau_ff"][i], out["tau_fb"][i] = ff, fb
        out["action"][i] = id_np.pd_target_action(ff + fb, q, v, kp, kd)
        out["state"][i] = id_np.policy_state(model, q, v, feet)
    return out


def _compare(got, ref):
    for k in ref:
        g = got[k].cpu().numpy()
        scale = np.maximum(1.0, np.abs(ref[k]).max(axis=1, keepdims=True))
        assert np.all(np.isfinite(g)), k
        assert (np.abs(g - ref[k]) / scale).max() < 1e-11, (k, (np.abs(g - ref[k]) / scale).max())


@pytest.mark.parametrize("robot", ["solo12", "go2"])
def test_rows_match_the_oracle(robot):
    import torch
    from bunmpc_amd import robot_id_controller as ric
    model = _setup(robot)
    s = _samples(model, 200, 21)
    kp, kd = np.linspace(2.0, 4.0, 12), np.linspace(0.05, 0.2, 12)
    ctrl = ric.InverseDynamicsController(model, FEET[robot])
    t = {k: torch.as_tensor(v, device="cuda:0") for k, v in s.items()}
    got = ric.id_batch_device(ctrl.dev_model, ctrl.foot_frames, kp, kd, t["q_des"], t["v_des"], t["a_des"], t["f"], t["q"], t["v"])
    _compare(got, _oracle(model, FEET[robot], s, kp, kd))
    # a sample exactly on its plan (no measured state given): no feedback, the state row is that of the desired state
    got = ric.id_batch_device(ctrl.dev_model, ctrl.foot_frames, kp, kd, t["q_des"], t["v_des"], t["a_des"], t["f"])
    ref = _oracle(model, FEET[robot], s, kp, kd, measured=False)
    assert np.all(got["tau_fb"].cpu().numpy() == 0)
    _compare(got, ref)


def test_plan_rows_in_place_and_ragged_sizes():
    """strided views of a 1 kHz plan (xs_int rows of 37), sizes around the 64-lane granularity, permuted foot order"""
    import torch
    from bunmpc_amd import robot_id_controller as ric
    model = _setup("solo12")
    feet = ["HR_FOOT", "FL_FOOT", "HL_FOOT", "FR_FOOT"]
    ctrl = ric.InverseDynamicsController(model, feet)
    ctrl.set_gains(3.0, 0.05)
    for n in (1, 63, 65):
        s = _samples(model, n, 30 + n)
        xs = torch.as_tensor(np.hstack([s["q_des"], s["v_des"]]), device="cuda:0")
        got = ctrl.rows(xs, torch.as_tensor(s["a_des"], device="cuda:0"), torch.as_tensor(s["f"], device="cuda:0"),
                        torch.as_tensor(s["q"], device="cuda:0"), torch.as_tensor(s["v"], device="cuda:0"))
        _compare(got, _oracle(model, feet, s, 3.0, 0.05))
    # empty batch: nothing launched, empty outputs
    e = torch.zeros((0, 37), dtype=torch.float64, device="cuda:0")
    got = ctrl.rows(e, e[:, :18], e[:, :12])
    assert got["action"].shape == (0, 12)


def test_end_effectors_on_any_body_of_a_leg():
    import torch
    from bunmpc_amd import robot_id_controller as ric
    model = _setup("solo12")
    feet = ["FL_UPPER_LEG", "FR_SHOULDER", "HL_FOOT", "HR_ANKLE"]
    ctrl = ric.InverseDynamicsController(model, feet)
    ctrl.set_gains(2.5, 0.1)
    s = _samples(model, 40, 5)
    t = {k: torch.as_tensor(v, device="cuda:0") for k, v in s.items()}
    got = ric.id_batch_device(ctrl.dev_model, ctrl.foot_frames, 2.5, 0.1, t["q_des"], t["v_des"], t["a_des"], t["f"], t["q"], t["v"])
    _compare(got, _oracle(model, feet, s, 2.5, 0.1))


def test_reference_class_surface():
    """InverseDynamicsController as the rollout loop calls it (simulation.py:512-524): one sample, numpy in and out"""
    from bunmpc_amd import robot_id_controller as ric
    from oracle import id_np
    model = _setup("solo12")
    s = _samples(model, 1, 77, unit_quat=True)
    ctrl = ric.InverseDynamicsController(model, FEET["solo12"])
    ctrl.set_gains(3.0, 0.05)
    tau, fb = ctrl.id_joint_torques(s["q"][0], s["v"][0], s["q_des"][0], s["v_des"][0], s["a_des"][0], s["f"][0])
    ref = id_np.InverseDynamicsController(model, FEET["solo12"])
    ref.set_gains(3.0, 0.05)
    rt, rf = ref.id_joint_torques(s["q"][0], s["v"][0], s["q_des"][0], s["v_des"][0], s["a_des"][0], s["f"][0])
    assert np.abs(tau - rt).max() < 1e-11 * max(1.0, np.abs(rt).max()) and np.abs(fb - rf).max() < 1e-13
    t = ctrl.compute_id_torques(s["q_des"][0], s["v_des"][0], s["a_des"][0])
    assert np.abs(t[6:] - id_np.rnea(model, s["q_des"][0], s["v_des"][0], s["a_des"][0])[6:]).max() < 1e-11 * np.abs(rt).max() + 1e-11


def test_argument_checks():
    import ctypes as C
    import torch
    from bunmpc_amd import _lib, robot_id_controller as ric
    model = _setup("solo12")
    ctrl = ric.InverseDynamicsController(model, FEET["solo12"])
    z = lambda w: torch.zeros((4, w), dtype=torch.float64, device="cuda:0")
    with pytest.raises(_lib.BmpcError, match="two end effectors on one leg"):
        ric.id_batch_device(ctrl.dev_model, [ctrl.foot_frames[0]] * 4, 1.0, 0.1, z(19), z(18), z(18), z(12))
    with pytest.raises(_lib.BmpcError, match="kp = 0"):
        ric.id_batch_device(ctrl.dev_model, ctrl.foot_frames, 0.0, 0.1, z(19), z(18), z(18), z(12))
    with pytest.raises(_lib.BmpcError, match="foot frame out of range"):
        ric.id_batch_device(ctrl.dev_model, [999, 1, 2, 3], 1.0, 0.1, z(19), z(18), z(18), z(12))
    d = _lib.IdBatch()
    assert _lib.lib().bmpc_id_batch_device(C.byref(d), None) == _lib.BAD_ARG

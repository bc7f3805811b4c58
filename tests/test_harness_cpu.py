"""CPU checks of the harness mirror's host arithmetic (bunmpc_amd/cyclic_gen.py, problems.py): the parts of
SoloMpcGaitGen that need no solve.  Hand-computed cases; the reference cannot be imported."""
import os

import numpy as np
import pytest

from bunmpc_amd import cyclic_gen, fk_np, problems, urdf_model

ROBOTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots")


@pytest.fixture(scope="module", params=["solo12", "go2"])
def robot(request):
    m = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, request.param + ".json")).read())
    return m, (problems.SOLO12_WB if request.param == "solo12" else problems.GO2_WB)


def test_interpolation_is_stacked_linspace_with_end_points():
    knots = np.array([[0.0, 10.0], [1.0, 20.0], [3.0, 0.0]])
    out = cyclic_gen.interpolate_plan(knots, np.array([0.05, 0.02]), 2)
    assert out.shape == (70, 2)
    assert np.array_equal(out[0], knots[0]) and np.array_equal(out[49], knots[1])      # end point included ...
    assert np.array_equal(out[50], knots[1]) and np.array_equal(out[69], knots[2])     # ... and repeated
    assert np.allclose(np.diff(out[:50, 0]), 1.0 / 49) and np.allclose(np.diff(out[50:, 0]), 2.0 / 19)


def test_go2_model_matches_the_xacro_constants():
    m = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "go2.json")).read())
    # const.xacro:70,82,94,106,119 + the 1 g base and imu links of go2.urdf.xacro:38,86
    assert abs(m.total_mass - (6.921 + 4 * (0.678 + 1.152 + 0.154 + 0.06) + 0.002)) < 1e-12
    assert m.joint_names == [l + j for l in ("FL", "FR", "RL", "RR") for j in ("_hip_joint", "_thigh_joint", "_calf_joint")]
    k = fk_np.kinematics(m, np.array([[0, 0, 0, 0, 0, 0, 1.0] + [0.0] * 12]))
    feet = fk_np.frame_positions(m, k, problems.GO2_WB.feet)[0]
    # straight legs: hip offset (0.1934, 0.0465) + thigh offset 0.0955 sideways, 2 x 0.213 down
    assert np.allclose(feet, [[0.1934, 0.142, -0.426], [0.1934, -0.142, -0.426], [-0.1934, 0.142, -0.426], [-0.1934, -0.142, -0.426]])
    k = fk_np.kinematics(m, problems.GO2_Q0[None])
    assert np.allclose(fk_np.frame_positions(m, k, problems.GO2_WB.feet)[0][:, 2], 0.02, atol=1e-4)


def test_composite_inertia_is_a_base_frame_quantity(robot):
    m, wb = robot
    I0 = cyclic_gen.composite_inertia_base(m, wb.q0)
    assert np.allclose(I0, I0.T) and np.all(np.linalg.eigvalsh(I0) > 0)
    q = wb.q0.copy()
    q[0:3] = [0.3, -0.2, 0.5]
    q[3:7] = np.array([0.1, -0.2, 0.3, 0.9]) / np.linalg.norm([0.1, -0.2, 0.3, 0.9])
    assert np.allclose(cyclic_gen.composite_inertia_base(m, q), I0, atol=1e-12)
    # a lower bound any body arrangement satisfies: at least the sum of the bodies' own inertias' traces
    assert np.trace(I0) > sum(np.trace(m.inertia[b]) for b in range(m.nj + 1))


def test_standing_robot_plan_by_hand(robot):
    """all feet in stance (stance 100 %), v_des = 0: every knot keeps the rounded current foot
    positions, X_nom holds the CoM xy and nom_ht, X_ter = [com_xy, nom_ht, 0...]"""
    import dataclasses
    m, wb = robot
    still = dataclasses.replace(problems.TROT, name="still", stance_percent=(1.0,) * 4, phase_offset=(0.0,) * 4)
    k = fk_np.kinematics(m, wb.q0[None], np.zeros((1, 18)))
    feet = np.round(fk_np.frame_positions(m, k, wb.feet), 3)
    rp = problems.RobotParams("r", m.total_mass, feet[0][:, :2], np.zeros((4, 2)), float(k["com"][0, 2]))
    cnt, swing, dt = problems.contact_plan(still, rp, 20, np.array([0.1]), np.round(k["com"][:, :2], 3), k["com"][:, 2],
                                           feet, np.zeros((1, 3)), np.zeros(1))
    assert np.all(cnt[0, :, :, 0] == 1) and np.all(swing == 0) and np.allclose(dt, 0.05)
    assert np.array_equal(cnt[0, :, :, 1:4], np.broadcast_to(feet[0], (20, 4, 3)))
    x_init = np.concatenate([k["com"], k["vcom"], k["L"]], axis=1)
    X_nom, X_ter = problems.centroidal_costs(still, 20, x_init, np.zeros((1, 3)), dt)
    X_nom = X_nom.reshape(20, 9)
    assert np.allclose(X_nom[:, 0], k["com"][0, 0]) and np.all(X_nom[:, 1] == 0) and np.all(X_nom[:, 2] == still.nom_ht)
    assert np.all(X_nom[:, 3:] == 0)
    assert np.allclose(X_ter[0], [k["com"][0, 0], k["com"][0, 1], still.nom_ht, 0, 0, 0, 0, 0, 0])


def test_first_knot_dt_rule():
    """abstract_cyclic_gen.py:385-388: dt_0 = gait_dt - round(t mod gait_dt, 2), or gait_dt when that is 0"""
    feet = np.zeros((3, 4, 3))
    _, _, dt = problems.contact_plan(problems.TROT, problems.SOLO12, 4, np.array([0.0, 0.02, 0.1]), np.zeros((3, 2)),
                                     np.full(3, 0.2), feet, np.zeros((3, 3)), np.zeros(3))
    assert np.allclose(dt[:, 0], [0.05, 0.03, 0.05]) and np.allclose(dt[:, 1:], 0.05)

"""CPU tests for the IK side: the numpy DDP oracle's own derivatives (finite differences), its
convergence on the reference's cost set, and the host-side behaviour of the InverseKinematics /
model C-ABI (no GPU).  PARITY UNPINNED vs crocoddyl/pinocchio (absent)."""
import os

import numpy as np
import pytest

from bunmpc_amd import _lib, problems, urdf_model
from oracle import ik_ddp_np, rbd_np as rb

ROBOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots", "solo12.json")
Q0 = problems.SOLO12_Q0


@pytest.fixture(scope="module")
def model():
    return urdf_model.RobotModel.from_json(open(ROBOT).read())


def trot_problem(model, T=4, seed=0):
    rng = np.random.default_rng(seed)
    wb = problems.make_wb_batch(model, 1, seed=77)
    names = list(model.frames)
    prob = ik_ddp_np.IKProblem(model, T)
    for t in range(T + 1):
        tk = wb.ik_tasks[0, min(t, wb.ik_T)]
        for s in range(4):
            if tk[5 * s] != 0:
                prob._add(t, "f%d" % s, ("frame", tk[5 * s], (names[int(tk[5 * s + 1])], tk[5 * s + 2:5 * s + 5])))
        prob._add(t, "com", ("com", 3.0, np.array([0.0, 0.0, 0.2]) + 0.01 * rng.standard_normal(3)))
        prob._add(t, "mom", ("mom", 5e2, 0.05 * rng.standard_normal(6)))
        prob._add(t, "x", ("state", 5e-2, (wb.state_w[0], wb.x_reg[0])))
        prob._add(t, "u", ("ctrl", 1e-5, wb.ctrl_w[0]))
    prob.setup_costs(np.full(T, 0.05))
    return prob, wb.x[0]


def test_node_derivatives_by_finite_differences(model):
    prob, x0 = trot_problem(model)
    rng = np.random.default_rng(1)
    x = rb.state_integrate(model, x0, 0.1 * rng.standard_normal(36))
    u = rng.standard_normal(18)
    d = ik_ddp_np.node_calc(prob, 1, x, u, diff=True)
    eps = 1e-6
    Lx_fd, Fx_fd = np.zeros(36), np.zeros((36, 36))
    for c in range(36):
        e = np.zeros(36); e[c] = eps
        dp = ik_ddp_np.node_calc(prob, 1, rb.state_integrate(model, x, e), u)
        dm = ik_ddp_np.node_calc(prob, 1, rb.state_integrate(model, x, -e), u)
        Lx_fd[c] = (dp["cost"] - dm["cost"]) / (2 * eps)
        Fx_fd[:, c] = rb.state_diff(model, dm["xnext"], dp["xnext"]) / (2 * eps)
    assert np.allclose(d["Lx"], Lx_fd, rtol=1e-5, atol=1e-6)
    assert np.allclose(d["Fx"], Fx_fd, atol=1e-6)
    Lu_fd, Fu_fd = np.zeros(18), np.zeros((36, 18))
    for c in range(18):
        e = np.zeros(18); e[c] = eps
        dp, dm = ik_ddp_np.node_calc(prob, 1, x, u + e), ik_ddp_np.node_calc(prob, 1, x, u - e)
        Lu_fd[c] = (dp["cost"] - dm["cost"]) / (2 * eps)
        Fu_fd[:, c] = rb.state_diff(model, dm["xnext"], dp["xnext"]) / (2 * eps)
    assert np.allclose(d["Lu"], Lu_fd, rtol=1e-5, atol=1e-8)
    assert np.allclose(d["Fu"], Fu_fd, atol=1e-6)
    # Gauss-Newton Hessian: symmetric PSD and equal to the true Hessian where residuals are linear (control)
    assert np.allclose(d["Lxx"], d["Lxx"].T) and np.linalg.eigvalsh(d["Lxx"]).min() > -1e-9
    assert np.allclose(np.diag(d["Luu"]), 0.05 * 1e-5 * problems.TROT_IK["ctrl_wt"])
    # terminal node: unscaled, no control term
    dT = ik_ddp_np.node_calc(prob, prob.T, x, None, diff=True)
    assert "xnext" not in dT and dT["cost"] > 0


def test_ddp_converges_like_gauss_newton(model):
    prob, x0 = trot_problem(model)
    r = ik_ddp_np.solve_ddp(prob, x0)
    assert r["converged"] and r["iters"] <= 12 and r["stop"] < 1e-9
    assert np.allclose(r["xs"][0], x0)                       # rollout starts at x0
    # the result is a rollout of the node model (no gaps left)
    for t in range(prob.T):
        nxt = ik_ddp_np.node_calc(prob, t, r["xs"][t], r["us"][t])["xnext"]
        assert np.allclose(nxt, r["xs"][t + 1], atol=1e-12)
    total = sum(ik_ddp_np.node_calc(prob, t, r["xs"][t], r["us"][t] if t < prob.T else None)["cost"] for t in range(prob.T + 1))
    assert total == pytest.approx(r["cost"], rel=1e-12)


def test_duplicate_cost_names_are_dropped_like_crocoddyl(model, capsys):
    prob = ik_ddp_np.IKProblem(model, 2)
    prob.add_com_position_tracking_task(0, 2, np.zeros((2, 3)), 1.0, "com_track", False)
    prob.add_com_position_tracking_task(0, 2, np.ones((2, 3)), 1.0, "com_track", False)
    assert "already existed" in capsys.readouterr().out
    assert np.all(prob.costs[0]["com_track"][2] == 0)


def _ik(model, n_col=3):
    from bunmpc_amd.inverse_kinematics_cpp import InverseKinematics
    return InverseKinematics(model, n_col)


def test_ik_host_side_argument_checking(model):
    ik = _ik(model)
    with pytest.raises(_lib.BmpcError):
        ik.add_position_tracking_task_single(999, np.zeros(3), 1.0, "x", 0)      # frame id out of range
    with pytest.raises(_lib.BmpcError):
        ik.add_position_tracking_task_single("FL_FOOT", np.zeros(3), 1.0, "x", 7)  # time step out of range
    with pytest.raises(ValueError):
        ik.add_state_regularization_cost(0, 3, 1.0, "xReg", np.ones(5), np.zeros(37), False)
    ik.add_state_regularization_cost(0, 3, 1.0, "xReg", np.ones(36), np.zeros(37), False)
    ik.add_state_regularization_cost_single(1, 1.0, "xReg", 2 * np.ones(36), np.zeros(37))   # duplicate name: dropped (warning)
    ik.add_state_regularization_cost_single(1, 1.0, "other", 2 * np.ones(36), np.zeros(37))  # a second state cost on node 1 ...
    with pytest.raises(_lib.BmpcError) as e:                                                 # ... is refused when the problem is packed
        ik.optimize(np.concatenate([Q0, np.zeros(18)]))
    assert e.value.code == 1
    ik.add_velocity_tracking_task(0, 0, 3, np.zeros(3), 1.0, "v")               # prints "function not implemented"
    with pytest.raises(_lib.BmpcError):
        ik.get_xs()                                                              # optimize not called yet
    assert model.frame_id("FL_FOOT") == 10 and model.frame_id("HR_FOOT") == 34   # pinocchio's frame order


def test_model_topology_is_validated(model):
    from bunmpc_amd.inverse_kinematics_cpp import DeviceModel
    bad = urdf_model.RobotModel.from_json(model.to_json())
    bad.parent = bad.parent.copy()
    bad.parent[4] = 0                      # branch below the base
    with pytest.raises(_lib.BmpcError):
        DeviceModel(bad)


def test_ik_optimize_needs_a_gpu(model):
    import ctypes as C
    n = C.c_int(0)
    rc = _lib.lib().bmpc_device_count(C.byref(n))
    if rc == _lib.OK and n.value > 0:
        pytest.skip("a GPU is present")
    ik = _ik(model)
    ik.setup_costs(np.full(3, 0.05))
    with pytest.raises(_lib.BmpcError) as e:
        ik.optimize(np.concatenate([Q0, np.zeros(18)]))
    assert e.value.code == _lib.DEVICE_ERROR

"""Finite-difference pins of the rigid-body restatement (oracle/rbd_np.py).  pinocchio is absent,
so every Jacobian is checked against the function it differentiates (PARITY UNPINNED vs pinocchio)."""
import os

import numpy as np
import pytest

from bunmpc_amd import urdf_model
from oracle import rbd_np as rb

ROBOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots", "solo12.json")


@pytest.fixture(scope="module")
def model():
    return urdf_model.RobotModel.from_json(open(ROBOT).read())


def rand_state(model, rng, scale=0.5):
    q = rb.integrate(model, rb.neutral(model), scale * rng.standard_normal(model.nv))
    q[7:] += np.tile([0.0, 0.8, -1.6], 4) * np.array([1] * 6 + [-1] * 6)
    return q, rng.standard_normal(model.nv)


def test_model_constants(model):
    assert model.nj == 12 and model.nq == 19 and model.nv == 18
    assert model.total_mass == pytest.approx(2.50000279, abs=1e-9)
    assert model.leg_chains() == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]]
    k = rb.Kin(model, np.array([0, 0, 0, 0, 0, 0, 1.0] + [0.0] * 12))
    # straight legs: foot under the hip chain, y = 0.0875 + 0.014 + 0.03745 + 0.008, z = -0.32
    assert np.allclose(k.frame_placement("FL_FOOT")[1], [0.1946, 0.14695, -0.32])
    assert np.allclose(k.frame_placement("HR_FOOT")[1], [-0.1946, -0.14695, -0.32])


def test_se3_log_exp_roundtrip_and_jacobians():
    rng = np.random.default_rng(0)
    for scale in (1e-9, 1e-3, 0.3, 2.0):
        nu = scale * rng.standard_normal(6)
        R, p = rb.exp6(nu)
        assert np.allclose(rb.log6(R, p), nu, atol=1e-10)
        J = rb.jlog6(R, p)
        Jfd = np.zeros((6, 6))
        eps = 1e-6
        for k in range(6):
            d = np.zeros(6); d[k] = eps
            dR, dp = rb.exp6(d)
            Jfd[:, k] = (rb.log6(R @ dR, R @ dp + p) - rb.log6(R, p)) / eps
        assert np.allclose(J, Jfd, atol=2e-5), scale
        # Jexp6: exp6(nu + d) = exp6(nu) exp6(Jexp6 d)
        Je = rb.jexp6(nu)
        for k in range(6):
            d = np.zeros(6); d[k] = eps
            R2, p2 = rb.exp6(nu + d)
            R3, p3 = rb.exp6(nu - d)
            fd = (rb.log6(R.T @ R2, R.T @ (p2 - p)) - rb.log6(R.T @ R3, R.T @ (p3 - p))) / (2 * eps)
            assert np.allclose(fd, Je[:, k], atol=1e-6)


def test_kinematic_jacobians_by_finite_differences(model):
    rng = np.random.default_rng(1)
    eps = 1e-6
    for _ in range(3):
        q, v = rand_state(model, rng)
        k0 = rb.Kin(model, q, v)
        Jf = {n: k0.frame_jacobian_lin(n) for n in ("FL_FOOT", "HR_FOOT", "FR_HFE")}
        Jc, Ag, dh = k0.jacobian_com(), k0.centroidal_map(), k0.dh_dq()
        h0 = k0.centroidal_momentum()
        assert np.allclose(Ag @ v, h0, atol=1e-12)                      # h_g = A_g v
        assert np.allclose(h0[:3], model.total_mass * (Jc @ v), atol=1e-12)   # linear part = m vcom
        for c in range(model.nv):
            d = np.zeros(model.nv); d[c] = eps
            k1 = rb.Kin(model, rb.integrate(model, q, d), v)
            for n, J in Jf.items():
                assert np.allclose((k1.frame_placement(n)[1] - k0.frame_placement(n)[1]) / eps, J[:, c], atol=5e-6), (n, c)
            assert np.allclose((k1.com - k0.com) / eps, Jc[:, c], atol=5e-6), c
            assert np.allclose((k1.centroidal_momentum() - h0) / eps, dh[:, c], atol=2e-5), c


def test_state_operators(model):
    rng = np.random.default_rng(2)
    q0, v0 = rand_state(model, rng)
    q1, v1 = rand_state(model, rng)
    x0, x1 = np.concatenate([q0, v0]), np.concatenate([q1, v1])
    d = rb.state_diff(model, x0, x1)
    back = rb.state_integrate(model, x0, d)
    assert np.allclose(rb.state_diff(model, back, x1), 0, atol=1e-9)     # x0 (+) (x1 (-) x0) = x1
    J = rb.state_jdiff_second(model, x0, x1)
    eps = 1e-6
    for c in range(2 * model.nv):
        e = np.zeros(2 * model.nv); e[c] = eps
        fd = (rb.state_diff(model, x0, rb.state_integrate(model, x1, e)) - d) / eps
        assert np.allclose(fd, J[:, c], atol=2e-5), c
    dx = 0.3 * rng.standard_normal(2 * model.nv)
    J1, J2 = rb.state_jintegrate(model, x0, dx)
    xn = rb.state_integrate(model, x0, dx)
    for c in range(2 * model.nv):
        e = np.zeros(2 * model.nv); e[c] = eps
        fd1 = rb.state_diff(model, xn, rb.state_integrate(model, rb.state_integrate(model, x0, e), dx)) / eps
        fd2 = rb.state_diff(model, xn, rb.state_integrate(model, x0, dx + e)) / eps
        assert np.allclose(fd1, J1[:, c], atol=2e-5), c
        assert np.allclose(fd2, J2[:, c], atol=2e-5), c

"""oracle/perturb_np.py: the properties the contact-conditioned perturbation must have (no reference vectors exist:
parity unpinned) -- stance feet do not move to first order, the projectors are projectors, rejection consumes draws in order."""
import os

import numpy as np

from bunmpc_amd import urdf_model
from oracle import perturb_np, rbd_np

ROBOTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots")
FEET = ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"]
MU, SIGMA = [0.0, 0.0, 0.0, 0.0], [0.05, 0.1, 0.2, 0.2]


def stance():
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "solo12.json")).read())
    q = rbd_np.neutral(model)
    q[2] = 0.24
    q[7:] = [0.0, 0.8, -1.6] * 4
    return model, q


def test_stance_feet_stay_and_velocity_projection_is_the_references():
    model, q = stance()
    rng = np.random.default_rng(2)
    for contact in ([1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 1, 1]):
        z = rng.normal(size=36)
        qn, vn, _ = perturb_np.candidate(model, FEET, q, np.zeros(18), contact, z, MU, SIGMA)
        J = perturb_np.contact_jacobian(model, q, FEET, contact)
        d = rbd_np.difference(model, q, qn)
        assert np.abs(J @ d).max() < 1e-14                       # tangent to the contact constraint
        pos, vel = perturb_np.spread(z, MU, SIGMA)
        assert np.abs((J * vel) @ vn).max() < 1e-13             # the reference projects `pos` with J * vel (elementwise)
        P = np.identity(18) - np.linalg.pinv(J * vel) @ (J * vel)
        assert np.allclose(vn, P @ pos, atol=1e-14) and np.allclose(P @ P, P, atol=1e-12)
        kin0, kin1 = rbd_np.Kin(model, q), rbd_np.Kin(model, qn)
        for n, c in zip(FEET, contact):
            if c:
                assert np.abs(kin1.frame_placement(n)[1] - kin0.frame_placement(n)[1]).max() < 5 * np.abs(d).max() ** 2


def test_flight_phase_is_unprojected_and_rejection_walks_the_draws():
    model, q = stance()
    rng = np.random.default_rng(3)
    z = rng.normal(size=(6, 36))
    qn, vn, k = perturb_np.sample(model, FEET, q, np.ones(18), [0, 0, 0, 0], z, MU, SIGMA)
    pos, vel = perturb_np.spread(z[k], MU, SIGMA)
    assert np.allclose(vn, 1.0 + vel) and np.allclose(rbd_np.difference(model, q, qn), pos, atol=1e-12)
    # feet on the ground plane: a draw is accepted only if no foot ends below z = 0
    q0 = q.copy()
    q0[2] -= min(rbd_np.Kin(model, q).frame_placement(n)[1][2] for n in FEET)
    qn, vn, k = perturb_np.sample(model, FEET, q0, np.zeros(18), [0, 0, 0, 0], z, MU, SIGMA)
    for j in range(len(z)):
        h = perturb_np.candidate(model, FEET, q0, np.zeros(18), [0, 0, 0, 0], z[j], MU, SIGMA)[2]
        assert (j == k) == (not np.any(h < 0)) or j > k >= 0
        if j == k:
            break
    assert k != 0 or not np.any(perturb_np.candidate(model, FEET, q0, np.zeros(18), [0, 0, 0, 0], z[0], MU, SIGMA)[2] < 0)

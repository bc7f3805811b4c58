"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the contact-conditioned perturbation sampler of the data
collection loop (ISL/examples/iterative_algorithm/data_collection.py:188-262; Jacobian rotation `utils.py:239-251`).

For a nominal state (q, v) and the feet the contact plan has on the ground:
  J          = stacked LOCAL_WORLD_ALIGNED linear Jacobians of those feet            (:204-214; rotate_jacobian(...)[0:3])
  pos        = mu_pos + sigma_pos * z_pos,   vel = mu_vel + sigma_vel * z_vel          (:232-236)
  d_pos      = (I - pinv(J) J) pos                                                      (:242-243)
  jac_vel    = Jdot * pos + J * vel   -- ELEMENTWISE, the vectors broadcast over the rows; Jdot = 0 here: the loop only
               calls computeJointJacobians, never computeJointJacobiansTimeVariation, so data.dJ keeps its zero
               initialisation and getFrameJacobianTimeVariation returns zeros      (:216-222, 244)
  d_vel      = (I - pinv(jac_vel) jac_vel) pos    -- `pos`, not `vel`: as written in the reference (:245-246)
  q' = integrate(q, d_pos),  v' = v + d_vel;  drawn again while any foot of q' is below the ground (:229, 249-259)
The normal draws are an input (z, standard normal), so that the GPU sampler and this file can be fed the same numbers;
the reference draws them from numpy's global generator.  Parity unpinned: no vectors in the reference; the projector
identities are checked in tests/test_perturb_cpu.py.
"""
import numpy as np

from . import rbd_np


def contact_jacobian(model, q, feet, contact):
    kin = rbd_np.Kin(model, q)
    rows = [kin.frame_jacobian_lin(n) for n, c in zip(feet, contact) if c == 1]
    return np.vstack(rows) if rows else np.zeros((0, model.nv))


def spread(z, mu, sigma):
    """z (36) standard normal -> (perturbation_pos, perturbation_vel); mu / sigma = (base pos, base ori, joint pos, vel)"""
    pos = np.concatenate([mu[0] + sigma[0] * z[0:3], mu[1] + sigma[1] * z[3:6], mu[2] + sigma[2] * z[6:18]])
    return pos, mu[3] + sigma[3] * z[18:36]


def candidate(model, feet, q, v, contact, z, mu, sigma):
    """one pass of the while loop: -> (q', v', foot heights)"""
    pos, vel = spread(np.asarray(z, float), mu, sigma)
    J = contact_jacobian(model, q, feet, contact)
    if J.shape[0] == 0:
        d_pos, d_vel = pos, vel
    else:
        n = model.nv
        d_pos = (np.identity(n) - np.linalg.pinv(J) @ J) @ pos
        jac_vel = np.zeros_like(J) * pos + J * vel
        d_vel = (np.identity(n) - np.linalg.pinv(jac_vel) @ jac_vel) @ pos
    qn = rbd_np.integrate(model, np.asarray(q, float), d_pos)
    kin = rbd_np.Kin(model, qn)
    return qn, np.asarray(v, float) + d_vel, np.array([kin.frame_placement(n)[1][2] for n in feet])


def sample(model, feet, q, v, contact, z, mu, sigma):
    """z (K, 36): the draws the loop would consume, in order -> (q', v', index of the accepted draw) or (None, None, -1)"""
    for k in range(len(z)):
        qn, vn, h = candidate(model, feet, q, v, contact, z[k], mu, sigma)
        if not np.any(h < 0.0):
            return qn, vn, k
    return None, None, -1

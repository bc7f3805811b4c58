"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's output stage --
`InverseDynamicsController` (ISL/examples/controllers/robot_id_controller.py:12-86) and the label / state rows the
rollout loop records from it (ISL/examples/iterative_algorithm/simulation.py:156-175, 484-528).

The arithmetic the reference takes from pinocchio 2.6.9 here is `pin.rnea` (recursive Newton-Euler in body frames,
free-flyer velocities in the base frame, gravity through the fictitious base acceleration) and
`pin.computeFrameJacobian(..., LOCAL_WORLD_ALIGNED)`; pinocchio is absent from this image, so this file restates the
published algorithm (Featherstone, RBDA table 5.1) in pinocchio's conventions -- including that the free-flyer's
rotation is Eigen's `quat.matrix()` of the quaternion AS GIVEN (no normalisation): the 1 kHz plan the controller is fed
interpolates states linearly (abstract_cyclic_gen.py:677-692), so its quaternions are not unit between knots, and the
base rotation enters only through gravity (R^T g) and through the world alignment of the foot forces (R^T f).
**Parity unpinned** (the reference holds no vectors for this path): `tests/test_id_cpu.py` pins it instead by an independent formulation -- Lagrange's equations
on finite differences of the kinetic / potential energy computed from `oracle/rbd_np.py`'s world-frame kinematics -- and
by the momentum-rate identity on the base rows.
"""
import numpy as np

from . import rbd_np

GRAVITY = 9.81


def quat_matrix(q):
    """Eigen::Quaternion::toRotationMatrix of (x, y, z, w), not normalised"""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _base_frame_kin(model, q):
    """kinematics of the limbs with the base at the identity (what the LOCAL Jacobians and placements depend on)"""
    qb = np.array(q, float)
    qb[:3] = 0.0
    qb[3:7] = [0.0, 0.0, 0.0, 1.0]
    return rbd_np.Kin(model, qb)


def _inertia_apply(m, c, Ic, mot):
    """spatial inertia (mass m, CoM c and rotational inertia Ic about the CoM, body frame) times a motion (lin, ang)"""
    lin = m * (mot[:3] + np.cross(mot[3:], c))
    return np.concatenate([lin, Ic @ mot[3:] + np.cross(c, lin)])


def _motion_cross(v, m):
    return np.concatenate([np.cross(v[3:], m[:3]) + np.cross(v[:3], m[3:]), np.cross(v[3:], m[3:])])


def _force_cross(v, f):
    return np.concatenate([np.cross(v[3:], f[:3]), np.cross(v[3:], f[3:]) + np.cross(v[:3], f[:3])])


def rnea(model, q, v, a):
    """pin.rnea(model, data, q, v, a) -> tau (nv).  robot_id_controller.py:48-56"""
    q, v, a = np.asarray(q, float), np.asarray(v, float), np.asarray(a, float)
    nj = model.nj
    Rb = quat_matrix(q[3:7])
    V = [v[:6].copy()]
    A = [a[:6] + np.concatenate([Rb.T @ np.array([0.0, 0.0, GRAVITY]), np.zeros(3)])]
    X = [None]
    for i in range(nj):
        b = model.parent[i] + 1
        R = model.R[i] @ rbd_np.exp3(model.axis[i] * q[7 + i])
        p = model.p[i]
        X.append((R, p))

        def act_inv(mot):
            return np.concatenate([R.T @ (mot[:3] - np.cross(p, mot[3:])), R.T @ mot[3:]])
        vJ = np.concatenate([np.zeros(3), model.axis[i] * v[6 + i]])
        vi = act_inv(V[b]) + vJ
        ai = act_inv(A[b]) + np.concatenate([np.zeros(3), model.axis[i] * a[6 + i]]) + _motion_cross(vi, vJ)
        V.append(vi)
        A.append(ai)
    F = []
    for b in range(nj + 1):
        h = _inertia_apply(model.mass[b], model.com[b], model.inertia[b], V[b])
        F.append(_inertia_apply(model.mass[b], model.com[b], model.inertia[b], A[b]) + _force_cross(V[b], h))
    tau = np.zeros(model.nv)
    for i in range(nj - 1, -1, -1):
        tau[6 + i] = model.axis[i] @ F[i + 1][3:]
        R, p = X[i + 1]
        lin = R @ F[i + 1][:3]
        F[model.parent[i] + 1] = F[model.parent[i] + 1] + np.concatenate([lin, R @ F[i + 1][3:] + np.cross(p, lin)])
    tau[:6] = F[0]
    return tau


class InverseDynamicsController:
    """robot_id_controller.py:12-86 with a RobotModel where the reference holds a pinocchio wrapper"""

    def __init__(self, model, eff_arr):
        self.model, self.eff_arr = model, list(eff_arr)
        self.nq, self.nv = model.nq, model.nv

    def set_gains(self, kp, kd):
        self.kp, self.kd = kp, kd

    def compute_id_torques(self, q, v, a):
        return rnea(self.model, q, v, a)

    def id_joint_torques(self, q, dq, des_q, des_v, des_a, fff):
        """:57-86 -> (tau, tau_gain), both over the actuated joints"""
        q, dq, des_q, des_v = (np.asarray(x, float) for x in (q, dq, des_q, des_v))
        tau_id = self.compute_id_torques(des_q, des_v, des_a)
        # J^T [f; 0] with J = [R_wf 0; 0 R_wf] J_LOCAL: the LOCAL Jacobian does not see the base rotation, the alignment
        # R_wf = Rb R_limb does, and its limb part cancels against the frame's own rotation
        kin, Rb = _base_frame_kin(self.model, des_q), quat_matrix(des_q[3:7])
        tau_eff = np.zeros(self.nv)
        for j, name in enumerate(self.eff_arr):
            tau_eff += kin.frame_jacobian_lin(name).T @ (Rb.T @ np.asarray(fff[3 * j:3 * j + 3], float))
        tau = (tau_id - tau_eff)[6:]
        tau_gain = -self.kp * (q[7:] - des_q[7:]) - self.kd * (dq[6:] - des_v[6:])
        return tau, tau_gain


def pd_target_action(tau, q, v, kp, kd):
    """simulation.py:523-524"""
    return (tau + kd * np.asarray(v)[6:]) / kp + np.asarray(q)[7:]


def policy_state(model, q, v, eff_arr):
    """the 43-entry state row: [v (18), base - foot in x, y per foot (8), q[2:] (17)]  (simulation.py:156-175, 489-491)"""
    q, v = np.asarray(q, float), np.asarray(v, float)
    kin, Rb = _base_frame_kin(model, q), quat_matrix(q[3:7])
    rel = np.concatenate([q[0:2] - (q[0:3] + Rb @ kin.frame_placement(n)[1])[0:2] for n in eff_arr])
    return np.concatenate([v, rel, q[2:]])


# ------------------------------------------------------------------ independent formulation used to pin rnea() ---
def mass_matrix(model, q):
    """M(q) = sum_b J_b^T I_b J_b from the world-frame body Jacobians of rbd_np.Kin"""
    kin = rbd_np.Kin(model, q)
    M = np.zeros((model.nv, model.nv))
    for b in range(model.nj + 1):
        cols = kin.support[b]
        S = kin.S[:, cols]
        IS = np.stack([kin._apply(kin.m[b], kin.cw[b], kin.Iw[b], S[:, k]) for k in range(len(cols))], axis=1)
        M[np.ix_(cols, cols)] += S.T @ IS
    return M


def kinetic_energy(model, q, v):
    return 0.5 * v @ mass_matrix(model, q) @ v


def potential_energy(model, q):
    kin = rbd_np.Kin(model, q)
    return kin.M * GRAVITY * kin.com[2]

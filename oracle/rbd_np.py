"""oracle/rbd_np.py -- numpy restatement of the rigid-body quantities the reference's IK takes from
pinocchio 2.6.9 (absent here): forward kinematics, frame / CoM Jacobians, centroidal momentum
h_g = A_g(q) v and its partial derivatives, SE(3) exp/log and their Jacobians, the
StateMultibody integrate / diff operators.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: pinocchio cannot be imported or built here, so these
follow pinocchio's documented conventions (free-flyer velocity in the base frame, quaternion
xyzw, centroidal momentum expressed at the CoM with world orientation, linear part first) and are
pinned by finite differences of their own definitions (tests/test_rbd_cpu.py), not by pinocchio
outputs.  Call sites in the reference: ISL/src/ik/action_model.cpp:60-63,82-86,
ISL/src/motion_planner/kino_dyn.cpp:42, ISL/src/ik/inverse_kinematics.cpp:77.
"""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


# ------------------------------------------------------------------ SO(3) / SE(3) ---
def exp3(w):
    t2 = w @ w
    t = np.sqrt(t2)
    if t < 1e-3:
        a, b = 1.0 - t2 / 6.0 + t2 * t2 / 120.0, 0.5 - t2 / 24.0 + t2 * t2 / 720.0
    else:
        a, b = np.sin(t) / t, (1.0 - np.cos(t)) / t2
    K = skew(w)
    return np.eye(3) + a * K + b * (K @ K)


def log3(R):
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    t = np.arctan2(0.5 * np.linalg.norm(v), (np.trace(R) - 1.0) / 2.0)   # accurate near 0, unlike arccos
    if t < 1e-3:
        return 0.5 * v * (1.0 + t * t / 6.0 + 7.0 * t ** 4 / 360.0)
    if np.pi - t < 1e-6:  # near pi: from the symmetric part
        A = (R + np.eye(3)) / 2.0
        k = np.argmax(np.diag(A))
        ax = A[:, k] / np.sqrt(A[k, k])
        if ax @ v < 0:
            ax = -ax
        return t * ax
    return t / (2.0 * np.sin(t)) * v


def exp6(nu):
    """nu = (v, w) -> (R, p): p = V(w) v"""
    v, w = nu[:3], nu[3:]
    t2 = w @ w
    t = np.sqrt(t2)
    if t < 1e-3:
        b, c = 0.5 - t2 / 24.0 + t2 * t2 / 720.0, 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0
    else:
        b, c = (1.0 - np.cos(t)) / t2, (t - np.sin(t)) / (t2 * t)
    K = skew(w)
    V = np.eye(3) + b * K + c * (K @ K)
    return exp3(w), V @ v


def log6(R, p):
    w = log3(R)
    t2 = w @ w
    t = np.sqrt(t2)
    K = skew(w)
    if t < 1e-3:
        beta = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0
    else:
        beta = 1.0 / t2 - np.sin(t) / (2.0 * t * (1.0 - np.cos(t)))
    Vinv = np.eye(3) - 0.5 * K + beta * (K @ K)
    return np.concatenate([Vinv @ p, w])


def jlog3(w):
    """d log3(R exp3(d)) / d d at d = 0 (right Jacobian of log)"""
    t2 = w @ w
    t = np.sqrt(t2)
    if t < 1e-3:
        alpha, diag = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0, 0.5 * (2.0 - t2 / 6.0 - t2 * t2 / 360.0)
    else:
        s1c = np.sin(t) / (1.0 - np.cos(t))
        alpha, diag = 1.0 / t2 - s1c / (2.0 * t), 0.5 * t * s1c
    return alpha * np.outer(w, w) + diag * np.eye(3) + 0.5 * skew(w)


def jexp3(w):
    """right Jacobian of exp3: exp3(w + d) = exp3(w) exp3(jexp3(w) d)"""
    t2 = w @ w
    t = np.sqrt(t2)
    if t < 1e-3:
        b, c = 0.5 - t2 / 24.0 + t2 * t2 / 720.0, 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0
    else:
        b, c = (1.0 - np.cos(t)) / t2, (t - np.sin(t)) / (t2 * t)
    K = skew(w)
    return np.eye(3) - b * K + c * (K @ K)


def _q_left(rho, phi):
    """Barfoot's Q block of the left SE(3) Jacobian for xi = (rho, phi)"""
    t2 = phi @ phi
    t = np.sqrt(t2)
    P, Rh = skew(phi), skew(rho)
    if t < 1e-2:
        c1, c2, c3 = (1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0, 1.0 / 24.0 - t2 / 720.0 + t2 * t2 / 40320.0,
                      1.0 / 120.0 - t2 / 2520.0 + t2 * t2 / 120960.0)
    else:
        st, ct = np.sin(t), np.cos(t)
        c1 = (t - st) / (t2 * t)
        c2 = (t2 + 2.0 * ct - 2.0) / (2.0 * t2 * t2)
        c3 = (2.0 * t - 3.0 * st + t * ct) / (2.0 * t2 * t2 * t)
    return (0.5 * Rh + c1 * (P @ Rh + Rh @ P + P @ Rh @ P)
            + c2 * (P @ P @ Rh + Rh @ P @ P - 3.0 * P @ Rh @ P)
            + c3 * (P @ Rh @ P @ P + P @ P @ Rh @ P))


def jexp6(nu):
    """right Jacobian of exp6: exp6(nu + d) = exp6(nu) exp6(jexp6(nu) d);  J_r(xi) = J_l(-xi)"""
    J = np.zeros((6, 6))
    Jr = jexp3(nu[3:])
    J[:3, :3] = Jr
    J[3:, 3:] = Jr
    J[:3, 3:] = _q_left(-nu[:3], -nu[3:])
    return J


def jlog6(R, p):
    """d log6(M exp6(d)) / d d at d = 0 = jexp6(log6(M))^-1 = [[A, -A Q A],[0, A]], A = jlog3"""
    nu = log6(R, p)
    A = jlog3(nu[3:])
    J = np.zeros((6, 6))
    J[:3, :3] = A
    J[3:, 3:] = A
    J[:3, 3:] = -A @ _q_left(-nu[:3], -nu[3:]) @ A
    return J


def act_inv_matrix(R, p):
    """6x6 action of M^{-1} on motions (lin, ang ordering)"""
    X = np.zeros((6, 6))
    X[:3, :3] = R.T
    X[3:, 3:] = R.T
    X[:3, 3:] = -R.T @ skew(p)
    return X


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


# ------------------------------------------------------------- configuration space ---
def integrate(model, q, dq):
    """pinocchio::integrate: free-flyer M <- M exp6(dq[:6]); joints additive"""
    R = quat_to_R(q[3:7] / np.linalg.norm(q[3:7]))
    dR, dp = exp6(dq[:6])
    out = np.array(q, dtype=float)
    out[:3] = q[:3] + R @ dp
    out[3:7] = R_to_quat(R @ dR)
    out[7:] = q[7:] + dq[6:]
    return out


def difference(model, q0, q1):
    """pinocchio::difference(q0, q1) = log6(M0^-1 M1) ++ (q1 - q0)_joints"""
    R0, R1 = quat_to_R(q0[3:7] / np.linalg.norm(q0[3:7])), quat_to_R(q1[3:7] / np.linalg.norm(q1[3:7]))
    return np.concatenate([log6(R0.T @ R1, R0.T @ (q1[:3] - q0[:3])), q1[7:] - q0[7:]])


def state_integrate(model, x, dx):
    nq, nv = model.nq, model.nv
    return np.concatenate([integrate(model, x[:nq], dx[:nv]), x[nq:] + dx[nv:]])


def state_diff(model, x0, x1):
    nq = model.nq
    return np.concatenate([difference(model, x0[:nq], x1[:nq]), x1[nq:] - x0[nq:]])


def state_jdiff_second(model, x0, x1):
    """d diff(x0, x1 (+) d)/dd : Jlog6 on the base block, identity elsewhere"""
    nq, nv = model.nq, model.nv
    R0, R1 = quat_to_R(x0[3:7] / np.linalg.norm(x0[3:7])), quat_to_R(x1[3:7] / np.linalg.norm(x1[3:7]))
    J = np.eye(2 * nv)
    J[:6, :6] = jlog6(R0.T @ R1, R0.T @ (x1[:3] - x0[:3]))
    return J


def state_jintegrate(model, x, dx):
    """(d/dx, d/d dx) of x (+) dx in tangent coordinates"""
    nv = model.nv
    R, p = exp6(dx[:6])
    J1, J2 = np.eye(2 * nv), np.eye(2 * nv)
    J1[:6, :6] = act_inv_matrix(R, p)
    J2[:6, :6] = jexp6(dx[:6])
    return J1, J2


def neutral(model):
    q = np.zeros(model.nq)
    q[6] = 1.0
    return q


# ---------------------------------------------------------------------- kinematics ---
class Kin:
    """World-frame kinematic quantities at (q, v).  Motion vectors are (v_O, w): velocity of the
    body-fixed point at the world origin, angular velocity.  Column k of S spans v[k]."""

    def __init__(self, model, q, v=None):
        nj, nv = model.nj, model.nv
        self.model = model
        Rb = quat_to_R(q[3:7] / np.linalg.norm(q[3:7]))
        self.oR = [Rb] + [None] * nj           # body 0 = base, body i+1 = joint i
        self.op = [np.array(q[:3], float)] + [None] * nj
        S = np.zeros((6, nv))
        for a in range(3):
            S[:3, a] = Rb[:, a]
            S[3:, 3 + a] = Rb[:, a]
            S[:3, 3 + a] = np.cross(self.op[0], Rb[:, a])
        for i in range(nj):
            b = model.parent[i] + 1
            Rp, pp = self.oR[b], self.op[b]
            Rj = Rp @ model.R[i] @ exp3(model.axis[i] * q[7 + i])
            pj = Rp @ model.p[i] + pp
            self.oR[i + 1], self.op[i + 1] = Rj, pj
            aw = Rj @ model.axis[i]
            S[3:, 6 + i] = aw
            S[:3, 6 + i] = np.cross(pj, aw)
        self.S = S
        # supports: which velocity columns move body b
        self.support = [list(range(6))]
        for i in range(nj):
            self.support.append(self.support[model.parent[i] + 1] + [6 + i])
        # per-body world inertia data
        self.m = model.mass
        self.cw = [self.oR[b] @ model.com[b] + self.op[b] for b in range(nj + 1)]
        self.Iw = [self.oR[b] @ model.inertia[b] @ self.oR[b].T for b in range(nj + 1)]
        self.M = float(self.m.sum())
        self.com = sum(self.m[b] * self.cw[b] for b in range(nj + 1)) / self.M
        self.v = None if v is None else np.asarray(v, float)
        if v is not None:
            self.V = [S[:, self.support[b]] @ self.v[self.support[b]] for b in range(nj + 1)]

    # subtree of velocity column k: bodies whose support contains k
    def subtree(self, k):
        return [b for b in range(self.model.nj + 1) if k in self.support[b]]

    def frame_placement(self, name):
        b, Rf, pf = self.model.frames[name]
        return self.oR[b] @ Rf, self.oR[b] @ pf + self.op[b]

    def frame_jacobian_lin(self, name):
        """LOCAL_WORLD_ALIGNED linear Jacobian of the frame origin (3 x nv)"""
        b, _, pf = self.model.frames[name]
        x = self.oR[b] @ pf + self.op[b]
        J = np.zeros((3, self.model.nv))
        for k in self.support[b]:
            J[:, k] = self.S[:3, k] + np.cross(self.S[3:, k], x)
        return J

    def _subtree_inertia(self, bodies):
        Ms = sum(self.m[b] for b in bodies)
        C = sum(self.m[b] * self.cw[b] for b in bodies) / Ms
        J = np.zeros((3, 3))
        for b in bodies:
            d = self.cw[b] - C
            J += self.Iw[b] + self.m[b] * ((d @ d) * np.eye(3) - np.outer(d, d))
        return Ms, C, J

    @staticmethod
    def _apply(Ms, C, J, mot):
        """composite inertia (about the world origin) times a motion -> (f, n_O)"""
        l = Ms * (mot[:3] + np.cross(mot[3:], C))
        return np.concatenate([l, J @ mot[3:] + np.cross(C, l)])

    def jacobian_com(self):
        J = np.zeros((3, self.model.nv))
        for k in range(self.model.nv):
            Ms, C, _ = self._subtree_inertia(self.subtree(k))
            J[:, k] = Ms / self.M * (self.S[:3, k] + np.cross(self.S[3:, k], C))
        return J

    def momentum_world(self):
        """(f, n_O) of the whole robot"""
        h = np.zeros(6)
        for b in range(self.model.nj + 1):
            h += self._apply(self.m[b], self.cw[b], self.Iw[b], self.V[b])
        return h

    def centroidal_momentum(self):
        """h_g = [m vcom ; L about the CoM], world orientation (pinocchio data.hg)"""
        h = self.momentum_world()
        return np.concatenate([h[:3], h[3:] - np.cross(self.com, h[:3])])

    def vcom(self):
        return self.momentum_world()[:3] / self.M

    def centroidal_map(self):
        """A_g (6 x nv) = dh_g/dv"""
        A = np.zeros((6, self.model.nv))
        for k in range(self.model.nv):
            h = self._apply(*self._subtree_inertia(self.subtree(k)), self.S[:, k])
            A[:, k] = np.concatenate([h[:3], h[3:] - np.cross(self.com, h[:3])])
        return A

    def dh_dq(self):
        """partial of h_g w.r.t. q (tangent coordinates), v held fixed.
        d h_O / d q_k = S_k x* h_sub(k) - I^c_k (S_k x V_parent(joint of k)); then re-centred at the
        moving CoM: n_g = n_O - c x f."""
        nv = self.model.nv
        hO = self.momentum_world()
        f = hO[:3]
        Jc = self.jacobian_com()
        out = np.zeros((6, nv))
        for k in range(nv):
            sub = self.subtree(k)
            hs = np.zeros(6)
            for b in sub:
                hs += self._apply(self.m[b], self.cw[b], self.Iw[b], self.V[b])
            Sk = self.S[:, k]
            # parent body velocity of the joint that owns column k
            if k < 6:
                Vp = np.zeros(6)
            else:
                Vp = self.V[self.model.parent[k - 6] + 1]
            cross_f = np.concatenate([np.cross(Sk[3:], hs[:3]), np.cross(Sk[3:], hs[3:]) + np.cross(Sk[:3], hs[:3])])
            SxV = np.concatenate([np.cross(Sk[3:], Vp[:3]) + np.cross(Sk[:3], Vp[3:]), np.cross(Sk[3:], Vp[3:])])
            dO = cross_f - self._apply(*self._subtree_inertia(sub), SxV)
            out[:3, k] = dO[:3]
            out[3:, k] = dO[3:] - np.cross(Jc[:, k], f) - np.cross(self.com, dO[:3])
        return out

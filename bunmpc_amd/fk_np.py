"""Batched (numpy, vectorised over the batch axis) forward kinematics for a RobotModel: what the
reference's harness asks pinocchio for while it builds the contact plan and the centroidal costs
(abstract_cyclic_gen.py:161-165,215,567-572): CoM, foot / hip frame positions, and the centroidal
state [com, vcom, L].  Host-side input generation only; the solver kernels have their own device
kinematics (csrc/rbd_device.h)."""
import numpy as np


def _quat_R(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _axis_angle_R(axis, ang):
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0.0]])
    s, c = np.sin(ang)[:, None, None], np.cos(ang)[:, None, None]
    return np.eye(3)[None] + s * K[None] + (1 - c) * (K @ K)[None]


def kinematics(model, q, v=None):
    """q (B,19), v (B,18) or None -> dict(oR (nb,B,3,3), op (nb,B,3), com (B,3), and with v: vcom, L)"""
    q = np.asarray(q, float)
    B, nj = q.shape[0], model.nj
    oR = [_quat_R(q[:, 3:7])]
    op = [q[:, 0:3].copy()]
    for i in range(nj):
        b = model.parent[i] + 1
        Rl = model.R[i][None] @ _axis_angle_R(model.axis[i], q[:, 7 + i])
        oR.append(oR[b] @ Rl)
        op.append(np.einsum("bij,j->bi", oR[b], model.p[i]) + op[b])
    cw = [np.einsum("bij,j->bi", oR[b], model.com[b]) + op[b] for b in range(nj + 1)]
    M = model.mass.sum()
    com = sum(model.mass[b] * cw[b] for b in range(nj + 1)) / M
    out = dict(oR=oR, op=op, com=com)
    if v is not None:
        v = np.asarray(v, float)
        # body twists (v_O, w) about the world origin
        w0 = np.einsum("bij,bj->bi", oR[0], v[:, 3:6])
        vl = np.einsum("bij,bj->bi", oR[0], v[:, 0:3])
        VO = [vl + np.cross(op[0], w0)]
        W = [w0]
        for i in range(nj):
            b = model.parent[i] + 1
            aw = np.einsum("bij,j->bi", oR[i + 1], model.axis[i]) * v[:, 6 + i:7 + i]
            W.append(W[b] + aw)
            VO.append(VO[b] + np.cross(op[i + 1], aw))
        f = np.zeros((B, 3))
        nO = np.zeros((B, 3))
        for b in range(nj + 1):
            vc = VO[b] + np.cross(W[b], cw[b])
            lb = model.mass[b] * vc
            Iw = oR[b] @ model.inertia[b][None] @ np.transpose(oR[b], (0, 2, 1))
            f += lb
            nO += np.einsum("bij,bj->bi", Iw, W[b]) + np.cross(cw[b], lb)
        out["vcom"] = f / M
        out["L"] = nO - np.cross(com, f)
    return out


def frame_positions(model, kin, names):
    """(B, len(names), 3)"""
    out = []
    for n in names:
        b, _, pf = model.frames[n]
        out.append(np.einsum("bij,j->bi", kin["oR"][b], pf) + kin["op"][b])
    return np.stack(out, axis=1)

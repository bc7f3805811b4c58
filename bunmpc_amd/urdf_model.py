"""URDF -> flat rigid-body model (what pinocchio::urdf::buildModel(urdf, JointModelFreeFlyer())
produces for the robots the reference uses, `ISL/src/ik/inverse_kinematics.cpp:10`,
`ISL/src/motion_planner/kino_dyn.cpp:9`), in plain numpy arrays that cross the C-ABI.

Conventions (pinocchio's): joint 0 = free-flyer "root_joint" carrying the URDF root link;
joints 1..n = the revolute joints in URDF traversal order; fixed joints are merged into the
supporting joint's body (inertias lumped, link/joint frames kept).  q = [p(3), quat xyzw(4),
joint angles], v = [v_lin(3) in the base frame, w(3) in the base frame, joint rates].
"""
import json
import xml.etree.ElementTree as ET

import numpy as np


def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(e):
    o = e.find("origin") if e is not None else None
    xyz = np.zeros(3) if o is None else np.array([float(t) for t in o.get("xyz", "0 0 0").split()])
    rpy = np.zeros(3) if o is None else np.array([float(t) for t in o.get("rpy", "0 0 0").split()])
    return rpy_to_R(*rpy), xyz


def _compose(a, b):
    """(R,p) of frame b expressed through frame a"""
    return a[0] @ b[0], a[0] @ b[1] + a[1]


def _add_inertia(body, m, c, I):
    """lump (mass m, com c, inertia I about c) -- all in the joint frame -- into body"""
    M = body["mass"] + m
    if M == 0:
        return
    cn = (body["mass"] * body["com"] + m * c) / M

    def shift(mass, com, In):
        d = com - cn
        return In + mass * ((d @ d) * np.eye(3) - np.outer(d, d))
    body["inertia"] = shift(body["mass"], body["com"], body["inertia"]) + shift(m, c, I)
    body["mass"], body["com"] = M, cn


class RobotModel:
    """nj movable joints after the free-flyer.  Arrays: parent (nj,) with -1 = base, placement
    R (nj,3,3) / p (nj,3) of each joint frame in its parent joint frame, axis (nj,3); bodies
    0..nj (0 = base): mass, com (joint frame), inertia about the com (joint frame);
    frames: name -> (body index, R, p)."""

    def __init__(self, joint_names, parent, R, p, axis, mass, com, inertia, frames, name="robot"):
        self.name = name
        self.joint_names = list(joint_names)
        self.parent = np.asarray(parent, dtype=np.int32)
        self.R, self.p, self.axis = np.asarray(R, float), np.asarray(p, float), np.asarray(axis, float)
        self.mass, self.com, self.inertia = np.asarray(mass, float), np.asarray(com, float), np.asarray(inertia, float)
        self.frames = frames
        self.nj = len(self.joint_names)
        self.nq, self.nv = 7 + self.nj, 6 + self.nj

    @property
    def total_mass(self):
        return float(self.mass.sum())

    def frame_id(self, name):
        """pinocchio-style integer id: position in the frame list"""
        return list(self.frames).index(name)

    def frame_by_id(self, fid):
        return list(self.frames.items())[fid]

    def to_json(self):
        return json.dumps(dict(
            name=self.name, joint_names=self.joint_names, parent=self.parent.tolist(), R=self.R.tolist(),
            p=self.p.tolist(), axis=self.axis.tolist(), mass=self.mass.tolist(), com=self.com.tolist(),
            inertia=self.inertia.tolist(),
            frames={k: [int(v[0]), np.asarray(v[1]).tolist(), np.asarray(v[2]).tolist()] for k, v in self.frames.items()}),
            indent=1)

    @staticmethod
    def from_json(text):
        d = json.loads(text)
        frames = {k: (v[0], np.array(v[1]), np.array(v[2])) for k, v in d["frames"].items()}
        return RobotModel(d["joint_names"], d["parent"], d["R"], d["p"], d["axis"], d["mass"], d["com"],
                          d["inertia"], frames, d.get("name", "robot"))

    def leg_chains(self):
        """The kernels handle a base with serial chains hanging off it: returns the list of chains
        (lists of joint indices, root first); raises if the tree has branches below the base."""
        children = {i: [j for j in range(self.nj) if self.parent[j] == i] for i in range(-1, self.nj)}
        chains = []
        for root in children[-1]:
            chain, cur = [root], root
            while children[cur]:
                if len(children[cur]) != 1:
                    raise ValueError("branching below the base is not supported")
                cur = children[cur][0]
                chain.append(cur)
            chains.append(chain)
        return chains


def load_urdf(path, name=None):
    root = ET.parse(path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    child_links = {j.find("child").get("link") for j in joints}
    roots = [n for n in links if n not in child_links]
    if len(roots) != 1:
        raise ValueError("URDF must have exactly one root link")
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.find("parent").get("link"), []).append(j)

    bodies = [dict(mass=0.0, com=np.zeros(3), inertia=np.zeros((3, 3)))]
    jn, parent, Rj, pj, axis = [], [], [], [], []
    frames = {"universe": (0, np.eye(3), np.zeros(3)), "root_joint": (0, np.eye(3), np.zeros(3))}

    def add_link(lname, body, T):
        """link frame T = (R,p) in the frame of joint `body` (0 = base)"""
        frames[lname] = (body, T[0], T[1])
        inn = links[lname].find("inertial")
        if inn is not None:
            Ri, pi = _origin(inn)
            m = float(inn.find("mass").get("value"))
            e = inn.find("inertia")
            g = {k: float(e.get(k, "0")) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")}
            I = np.array([[g["ixx"], g["ixy"], g["ixz"]], [g["ixy"], g["iyy"], g["iyz"]], [g["ixz"], g["iyz"], g["izz"]]])
            Rc, pc = _compose(T, (Ri, pi))
            _add_inertia(bodies[body], m, pc, Rc @ I @ Rc.T)
        # urdfdom keeps joints in a name-sorted map, so pinocchio visits children in that order
        for j in sorted(by_parent.get(lname, []), key=lambda e: e.get("name")):
            Tj = _compose(T, _origin(j))
            jtype = j.get("type")
            cl = j.find("child").get("link")
            if jtype == "fixed":
                frames[j.get("name")] = (body, Tj[0], Tj[1])
                add_link(cl, body, Tj)
            elif jtype in ("revolute", "continuous"):
                ax = j.find("axis")
                a = np.array([1.0, 0, 0]) if ax is None else np.array([float(t) for t in ax.get("xyz").split()])
                jn.append(j.get("name"))
                parent.append(body - 1)
                Rj.append(Tj[0]); pj.append(Tj[1]); axis.append(a / np.linalg.norm(a))
                bodies.append(dict(mass=0.0, com=np.zeros(3), inertia=np.zeros((3, 3))))
                nb = len(bodies) - 1
                frames[j.get("name")] = (nb, np.eye(3), np.zeros(3))
                add_link(cl, nb, (np.eye(3), np.zeros(3)))
            else:
                raise ValueError("unsupported joint type " + str(jtype))

    add_link(roots[0], 0, (np.eye(3), np.zeros(3)))
    return RobotModel(jn, parent, Rj, pj, axis, [b["mass"] for b in bodies], [b["com"] for b in bodies],
                      [b["inertia"] for b in bodies], frames, name or root.get("name", "robot"))

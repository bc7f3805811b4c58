"""Writes bunmpc_amd/robots/go2.json: the Go2 rigid-body model (free-flyer + 4 legs x 3 revolute).

The reference ships the Go2 only as xacro macros (robot_properties_go2/src/robot_properties_go2/
resources/xacro/{const,leg,go2.urdf}.xacro) and xacro is not installed here, so the macro is
expanded by hand below: the numbers are the <xacro:property> values of const.xacro (line cited next
to each), the tree is the one go2.urdf.xacro:31-132 + leg.xacro:7-172 describe.  The URDF text is
built in memory and pushed through bunmpc_amd.urdf_model.load_urdf, i.e. the same path Solo12's URDF takes.

    python tools/make_go2_model.py
"""
import io
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bunmpc_amd.urdf_model import load_urdf  # noqa: E402

# const.xacro
THIGH_OFFSET = 0.0955            # :26
THIGH_LEN = CALF_LEN = 0.213     # :27-28
LEG_X, LEG_Y = 0.1934, 0.0465    # :31-32
FOOT_R = 0.02                    # :21
TRUNK = dict(m=6.921, c=(0.021112, 0.0, -0.005366),                                    # :70-73
             I=(0.02448, 0.00012166, 0.0014849, 0.098077, -3.12e-05, 0.107))           # :74-79
HIP = dict(m=0.678, c=(-0.0054, 0.00194, -0.000105),                                   # :82-85
           I=(0.00048, -3.01e-06, 1.11e-06, 0.000884, -1.42e-06, 0.000596))            # :86-91
THIGH = dict(m=1.152, c=(-0.00374, -0.0223, -0.0327),                                  # :94-97
             I=(0.00584, 8.72e-05, -0.000289, 0.0058, 0.000808, 0.00103))              # :98-103
CALF = dict(m=0.154, c=(0.00548, -0.000975, -0.115),                                   # :106-109
            I=(0.00108, 3.4e-07, 1.72e-05, 0.0011, 8.28e-06, 3.29e-05))                # :110-115
FOOT_M = 0.06                    # :119


def link(name, m, c, I):
    ixx, ixy, ixz, iyy, iyz, izz = I
    return (f'<link name="{name}"><inertial><origin rpy="0 0 0" xyz="{c[0]!r} {c[1]!r} {c[2]!r}"/>'
            f'<mass value="{m!r}"/><inertia ixx="{ixx!r}" ixy="{ixy!r}" ixz="{ixz!r}" iyy="{iyy!r}" '
            f'iyz="{iyz!r}" izz="{izz!r}"/></inertial></link>\n')


def joint(name, kind, parent, child, xyz, axis=None):
    ax = "" if axis is None else f'<axis xyz="{axis}"/>'
    return (f'<joint name="{name}" type="{kind}"><origin rpy="0 0 0" xyz="{xyz[0]!r} {xyz[1]!r} {xyz[2]!r}"/>'
            f'<parent link="{parent}"/><child link="{child}"/>{ax}</joint>\n')


def leg(name, mirror, front):
    """leg.xacro:7-172 with mirror = +1 left / -1 right, front = +1 front / -1 hind"""
    s = joint(f"{name}_hip_joint", "revolute", "trunk", f"{name}_hip", (front * LEG_X, mirror * LEG_Y, 0.0), "1 0 0")
    ixx, ixy, ixz, iyy, iyz, izz = HIP["I"]
    s += link(f"{name}_hip", HIP["m"], (HIP["c"][0] * front, HIP["c"][1] * mirror, HIP["c"][2]),
              (ixx, ixy * mirror * front, ixz * front, iyy, iyz * mirror, izz))                     # :60-66
    s += joint(f"{name}_thigh_joint", "revolute", f"{name}_hip", f"{name}_thigh", (0.0, THIGH_OFFSET * mirror, 0.0), "0 1 0")
    ixx, ixy, ixz, iyy, iyz, izz = THIGH["I"]
    s += link(f"{name}_thigh", THIGH["m"], (THIGH["c"][0], THIGH["c"][1] * mirror, THIGH["c"][2]),
              (ixx, ixy * mirror, ixz, iyy, iyz * mirror, izz))                                     # :98-104
    s += joint(f"{name}_calf_joint", "revolute", f"{name}_thigh", f"{name}_calf", (0.0, 0.0, -THIGH_LEN), "0 1 0")
    s += link(f"{name}_calf", CALF["m"], CALF["c"], CALF["I"])                                      # :132-138
    s += joint(f"{name}_foot_joint", "fixed", f"{name}_calf", f"{name}_foot", (0.0, 0.0, -CALF_LEN))
    fi = 2 * FOOT_M / 5.0 * FOOT_R * FOOT_R                                                         # :162-166
    s += link(f"{name}_foot", FOOT_M, (0.0, 0.0, 0.0), (fi, 0.0, 0.0, fi, 0.0, fi))
    return s


def go2_urdf():
    s = '<?xml version="1.0"?>\n<robot name="go2">\n'
    s += link("base", 0.001, TRUNK["c"], TRUNK["I"])                      # go2.urdf.xacro:31-46
    s += joint("floating_base", "fixed", "base", "trunk", (0.0, 0.0, 0.0))
    s += link("trunk", TRUNK["m"], TRUNK["c"], TRUNK["I"])                # :54-76
    s += joint("imu_joint", "fixed", "trunk", "imu_link", (0.0, 0.0, 0.0))
    s += link("imu_link", 0.001, (0.0, 0.0, 0.0), (0.0001, 0.0, 0.0, 0.0001, 0.0, 0.0001))   # :84-89
    for name, mirror, front in (("FR", -1, 1), ("FL", 1, 1), ("RR", -1, -1), ("RL", 1, -1)):  # :128-131
        s += leg(name, mirror, front)
    return s + "</robot>\n"


if __name__ == "__main__":
    model = load_urdf(io.StringIO(go2_urdf()), "go2")
    out = os.path.join(os.path.dirname(__file__), "..", "bunmpc_amd", "robots", "go2.json")
    with open(out, "w") as f:
        f.write(model.to_json())
    print(out, "mass", model.total_mass, "joints", model.joint_names)

"""Synthetic MPC problem batches at the centroidal level.

Host-side numpy restatement of the *inputs* the reference's Python harness hands to
`BiconvexMP` (contact plan, dt, X_nom, X_ter, bounds, weights), vectorised over the
batch axis.  Follows, without importing it,
  iterative_supervised_learning/examples/mpc/abstract_cyclic_gen.py:159-414  (create_cnt_plan)
  iterative_supervised_learning/examples/mpc/abstract_cyclic_gen.py:532-614  (create_costs)
  iterative_supervised_learning/examples/motions/cyclic/solo12_{trot,bound,jump}.py (gait constants)
at the centroidal level: the robot state enters only through CoM / foot positions
(no pinocchio), exactly as SURVEY.md 8d defines configs 1-4.  Go2 and "pace" do not
exist in the reference: those sets are synthetic (documented in DESIGN.md).
"""
from dataclasses import dataclass, field, replace

import numpy as np

GRAVITY = 9.81
FOOT_SIZE = 0.018          # abstract_cyclic_gen.py:31
BOUNDS_TILE = np.array([-0.45, -0.45, 0.0, 0.45, 0.45, 0.45])  # :92-97,611
BASE_SEED = 20250202


@dataclass
class GaitParams:
    """BiconvexMotionParams subset needed by the centroidal solve (weight_abstract.py:7-42)."""
    name: str
    gait_period: float
    stance_percent: tuple
    phase_offset: tuple
    gait_horizon: float
    W_X: np.ndarray
    W_X_ter: np.ndarray
    W_F: np.ndarray            # (12,) one knot: 4 feet x 3
    nom_ht: float
    rho: float = 5e4
    gait_dt: float = 0.05
    step_ht: float = 0.075
    ori_correction: tuple = (0.3, 0.5, 0.4)

    @property
    def horizon(self):
        # abstract_cyclic_gen.py:125
        return int(np.round(self.gait_horizon * self.gait_period / self.gait_dt, 2))


@dataclass
class RobotParams:
    name: str
    mass: float
    feet_xy: np.ndarray        # (4,2) nominal foot xy (FL,FR,HL,HR)
    offsets_xy: np.ndarray     # (4,2) hip-minus-com offsets after rounding and +-0.04 widening
    com_height: float


# solo12.urdf link masses summed; feet under the hips
SOLO12 = RobotParams(
    "solo12", 2.50000279,
    np.array([[0.1946, 0.14695], [0.1946, -0.14695], [-0.1946, 0.14695], [-0.1946, -0.14695]]),
    np.array([[0.195, 0.142], [0.195, -0.142], [-0.195, 0.142], [-0.195, -0.142]]),
    0.23)
# synthetic Go2 (const.xacro masses/geometry)
GO2 = RobotParams(
    "go2", 15.097,
    np.array([[0.1934, 0.142], [0.1934, -0.142], [-0.1934, 0.142], [-0.1934, -0.142]]),
    np.array([[0.193, 0.182], [0.193, -0.182], [-0.193, 0.182], [-0.193, -0.182]]),
    0.30)

# motions/cyclic/solo12_trot.py:16-41
TROT = GaitParams(
    "trot", 0.5, (0.6,) * 4, (0.0, 0.5, 0.5, 0.0), 2.0,
    np.array([1e-5, 1e-5, 1e5, 1e1, 1e1, 2e2, 1e4, 1e4, 1e4]),
    10 * np.array([1e5, 1e-5, 1e5, 1e1, 1e1, 2e2, 1e5, 1e5, 1e5]),
    np.array(4 * [1e1, 1e1, 1e1]), 0.2, step_ht=0.075, ori_correction=(0.3, 0.5, 0.4))
# motions/cyclic/solo12_bound.py:16-44
BOUND = GaitParams(
    "bound", 0.3, (0.5,) * 4, (0.0, 0.0, 0.5, 0.5), 4.0,
    np.array([1e-5, 1e-5, 5e4, 1e1, 1e1, 1e3, 5e3, 1e4, 5e3]),
    10 * np.array([1e-5, 1e-5, 5e4, 1e1, 1e1, 1e3, 1e4, 1e4, 1e4]),
    np.array(4 * [1e1, 1e1, 1.5e1]), 0.25, step_ht=0.07, ori_correction=(0.2, 0.8, 0.8))
# motions/cyclic/solo12_jump.py:17-43
JUMP = GaitParams(
    "jump", 0.5, (0.3,) * 4, (0.7, 0.7, 0.7, 0.7), 3.0,
    TROT.W_X.copy(), TROT.W_X_ter.copy(), np.array(4 * [1e1, 1e1, 1.5e1]), 0.25,
    step_ht=0.05, ori_correction=(0.2, 0.5, 0.4))
# synthetic: trot weights, lateral pairs in phase
PACE = GaitParams(
    "pace", 0.5, (0.6,) * 4, (0.0, 0.5, 0.0, 0.5), 2.0,
    TROT.W_X.copy(), TROT.W_X_ter.copy(), TROT.W_F.copy(), 0.2)
GAITS = {"trot": TROT, "bound": BOUND, "jump": JUMP, "pace": PACE}


# ---------------------------------------------------------------- gait phase ---
def gait_phi(t, period, offset):
    """gait_planner.cpp:41-44 (vectorised)."""
    return np.fmod(t + offset * period, period)


def gait_phase(t, period, stance_percent, offset):
    """gait_planner.cpp:46-58, scalar overload incl. the 1e-4 slack."""
    st = period * stance_percent
    phi = gait_phi(t, period, offset)
    return ((phi <= st) | (np.abs(phi - st) < 1e-4)).astype(np.float64)


def gait_percent_in_phase(t, period, stance_percent, offset):
    """gait_planner.cpp:112-128."""
    st = period * stance_percent
    phi = gait_phi(t, period, offset)
    with np.errstate(divide="ignore", invalid="ignore"):     # 100 % stance: the swing branch is never selected
        return np.where(phi <= st, phi / st, (phi - st) / (period - st))


# ------------------------------------------------------------- contact plan ---
def contact_plan(gait, robot, H, t0, com_xy, z_height, feet0, v_des, w_des, hip_offsets=None):
    """create_cnt_plan (abstract_cyclic_gen.py:159-414), data path (no MCTS / height map /
    noise), vectorised over the batch.
      t0 (B,), com_xy (B,2) already rounded, z_height (B,), feet0 (B,4,3) already rounded,
      v_des (B,3) in the yaw frame (yaw = 0 here, R = I), w_des (B,)
    returns cnt_plan (B,H,4,4), swing_time (B,H,4), dt (B,H)."""
    B = t0.shape[0]
    E = 4
    gdt = gait.gait_dt
    cnt = np.zeros((B, H, E, 4))
    swing = np.zeros((B, H, E))
    vtrack = v_des[:, 0:2]
    # np.cross(ang_step(2-vector), [0,0,w]) -> (a_y w, -a_x w, 0)   (:285-286)
    ang2 = 0.5 * np.sqrt(z_height / GRAVITY)[:, None] * vtrack
    ang_step = np.stack([ang2[:, 1] * w_des, -ang2[:, 0] * w_des], axis=1)
    for i in range(H):
        for j in range(E):
            sp, off = gait.stance_percent[j], gait.phase_offset[j]
            if i == 0:
                ph = gait_phase(t0, gait.gait_period, sp, off)
                cnt[:, 0, j, 0] = ph
                cnt[:, 0, j, 1:4] = feet0[:, j]
                continue
            ft = np.round(t0 + i * gdt, 3)
            ph = gait_phase(ft, gait.gait_period, sp, off)
            hoff = robot.offsets_xy[j][None, :] if hip_offsets is None else hip_offsets[:, j]   # (R_yaw offsets[j])[0:2]
            hip_loc = com_xy + hoff + i * gdt * vtrack
            raibert = 0.5 * vtrack * gait.gait_period * sp - 0.05 * (vtrack - v_des[:, 0:2])
            per_ph = np.round(gait_percent_in_phase(ft, gait.gait_period, sp, off), 3)
            prev_on = cnt[:, i - 1, j, 0] == 1
            # stance, continuing contact: copy; stance, new contact: raibert + hip + ang
            new_xy = raibert + hip_loc + ang_step
            sw_xy = np.where((per_ph < 0.5)[:, None], hip_loc + ang_step,
                             hip_loc + ang_step + raibert)
            on = ph == 1
            xy = np.where(on[:, None], np.where(prev_on[:, None], cnt[:, i - 1, j, 1:3], new_xy),
                          sw_xy)
            z = np.where(on & prev_on, cnt[:, i - 1, j, 3], FOOT_SIZE)
            cnt[:, i, j, 0] = ph
            cnt[:, i, j, 1:3] = xy
            cnt[:, i, j, 3] = z
            swing[:, i, j] = np.where(~on & (per_ph - 0.5 < 0.02), 1.0, 0.0)
    dt = np.full((B, H), gdt)
    dt0 = gdt - np.round(np.remainder(t0, gdt), 2)   # :385-388
    dt[:, 0] = np.where(dt0 == 0, gdt, dt0)
    return cnt, swing, dt


def centroidal_costs(gait, H, x_init, v_des, dt, amom=None):
    """create_costs, dynamics part (abstract_cyclic_gen.py:564-607), w_des = 0 branch unless
    yaw momentum is supplied through amom.  Returns X_nom (B,9H), X_ter (B,9)."""
    B = x_init.shape[0]
    if amom is None:
        amom = np.zeros((B, 3))
    X_nom = np.zeros((B, H, 9))
    X_nom[:, :, 0] = x_init[:, 0:1]
    for i in range(1, H):
        X_nom[:, i, 0] = X_nom[:, i - 1, 0] + v_des[:, 0] * dt[:, i]
        X_nom[:, i, 1] = X_nom[:, i - 1, 1] + v_des[:, 1] * dt[:, i]
    X_nom[:, :, 2] = gait.nom_ht
    X_nom[:, :, 3:6] = v_des[:, None, :]
    X_nom[:, :, 6:9] = (amom * np.asarray(gait.ori_correction))[:, None, :]
    X_ter = np.zeros((B, 9))
    X_ter[:, 0:2] = x_init[:, 0:2] + gait.gait_horizon * gait.gait_period * v_des[:, 0:2]
    X_ter[:, 2] = gait.nom_ht
    X_ter[:, 3:6] = v_des
    X_ter[:, 6:9] = amom
    return X_nom.reshape(B, 9 * H), X_ter


@dataclass
class Batch:
    """Inputs of B independent BiconvexMP solves, harness level."""
    name: str
    B: int
    H: int
    E: int
    m: float
    rho: float
    cnt_plan: np.ndarray       # (B,H,E,4)
    dt: np.ndarray             # (B,H)
    x_init: np.ndarray         # (B,9)
    X_nom: np.ndarray          # (B,9H)
    X_ter: np.ndarray          # (B,9)
    W_X: np.ndarray            # (1 or B, 9H)   tile(W_X, H)
    W_X_ter: np.ndarray        # (1 or B, 9)
    W_F: np.ndarray            # (1 or B, 3EH)  tile(W_F, H)
    bounds: np.ndarray         # (1 or B, H, 6)
    swing_time: np.ndarray = None
    gait_id: np.ndarray = None
    mu: float = 1.0            # friction coefficient of the "SoC" projection (fista.hpp:60)
    meta: dict = field(default_factory=dict)

    def warm_start(self):
        """KinoDynMP::set_warm_starts (kino_dyn.cpp:83-99): X = tile(x0), F = 0, P = 0."""
        X = np.tile(self.x_init, (1, self.H + 1))
        F = np.zeros((self.B, 3 * self.E * self.H))
        P = np.zeros((self.B, 9 * (self.H + 1)))
        return X, F, P

    def take(self, idx):
        """sub-batch of the problems listed in idx"""
        idx = np.asarray(idx)

        def s(a):
            return a if a is None or a.shape[0] == 1 else a[idx]
        return Batch(self.name, len(idx), self.H, self.E, self.m, self.rho, self.cnt_plan[idx],
                     self.dt[idx], self.x_init[idx], self.X_nom[idx], self.X_ter[idx],
                     s(self.W_X), s(self.W_X_ter), s(self.W_F), s(self.bounds),
                     None if self.swing_time is None else self.swing_time[idx],
                     None if self.gait_id is None else self.gait_id[idx], self.mu, dict(self.meta))

    def slice(self, lo, hi):
        def s(a):
            return a if a is None or a.shape[0] == 1 else a[lo:hi]
        return Batch(self.name, hi - lo, self.H, self.E, self.m, self.rho, self.cnt_plan[lo:hi],
                     self.dt[lo:hi], self.x_init[lo:hi], self.X_nom[lo:hi], self.X_ter[lo:hi],
                     s(self.W_X), s(self.W_X_ter), s(self.W_F), s(self.bounds),
                     None if self.swing_time is None else self.swing_time[lo:hi],
                     None if self.gait_id is None else self.gait_id[lo:hi], self.mu, dict(self.meta))


def _draws(seed, first, B, n):
    """Per-problem independent streams: problem b always sees the same numbers whatever the
    batch size or the rank that generates it (SeedSequence.spawn keyed by absolute index)."""
    out_u = np.empty((B, n))
    out_n = np.empty((B, n))
    for b in range(B):
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, first + b])))
        out_u[b] = g.random(n)
        out_n[b] = g.standard_normal(n)
    return out_u, out_n


def make_batch(config, B, first=0, seed=None, H=None):
    """SURVEY.md 8d configs.
      "solo12_trot_nominal"  config 1 (every problem identical, B usually 1)
      "solo12_trot"          config 2 perturbed ICs
      "go2_bound"            config 3 (synthetic robot), H forced to 40, nom_ht 0.30, mu = 10:
                             with the reference's fixed mu = 1 its squared-norm "SoC" projection
                             blows up for a 15 kg robot (tests/test_oracle_cpu.py) -- the one
                             parameter a Go2 user would have to change (set_friction_coefficient,
                             biconvex.hpp:131)
      "solo12_mixed"         config 4 trot/bound/pace with per-problem weights, H = 20
    `first` = absolute index of problem 0 (for rank sharding)."""
    cfg_index = {"solo12_trot_nominal": 1, "solo12_trot": 2, "go2_bound": 3, "solo12_mixed": 4}[config]
    seed = BASE_SEED + cfg_index if seed is None else seed
    robot = GO2 if config == "go2_bound" else SOLO12
    E = 4
    if config == "solo12_mixed":
        gaits = [TROT, BOUND, PACE]
        H = 20 if H is None else H
    elif config == "go2_bound":
        gaits = [replace(BOUND, nom_ht=0.30)]
        H = 40 if H is None else H
    else:
        gaits = [TROT]
        H = TROT.horizon if H is None else H
    u, nrm = _draws(seed, first, B, 32)
    x_init = np.zeros((B, 9))
    x_init[:, 2] = robot.com_height
    v_des = np.zeros((B, 3))
    feet0 = np.zeros((B, E, 3))
    feet0[:, :, 0:2] = robot.feet_xy[None]
    feet0[:, :, 2] = FOOT_SIZE
    if config == "solo12_trot_nominal":
        t0 = np.zeros(B)
        v_des[:, 0] = 0.2
        gid = np.zeros(B, dtype=np.int64)
    else:
        t0 = np.round(0.05 * np.floor(u[:, 0] * 10), 3)
        v_des[:, 0] = 0.3 * u[:, 1]
        gid = np.minimum((u[:, 2] * len(gaits)).astype(np.int64), len(gaits) - 1)
        x_init[:, 0:3] += nrm[:, 0:3] * np.array([0.02, 0.02, 0.01])
        x_init[:, 3:6] += nrm[:, 3:6] * 0.1
        x_init[:, 6:9] += nrm[:, 6:9] * 0.02
        feet0[:, :, 0:2] += 0.01 * nrm[:, 9:17].reshape(B, E, 2)
    feet0_raw = feet0.copy()
    feet0 = np.round(feet0, 3)                       # :215 np.round(oMf.translation, 3)
    com_xy = np.round(x_init[:, 0:2], 3)             # :164
    w_des = np.zeros(B)
    cnt = np.zeros((B, H, E, 4))
    swing = np.zeros((B, H, E))
    dt = np.zeros((B, H))
    X_nom = np.zeros((B, 9 * H))
    X_ter = np.zeros((B, 9))
    for k, g in enumerate(gaits):
        sel = np.nonzero(gid == k)[0]
        if sel.size == 0:
            continue
        c, s, d = contact_plan(g, robot, H, t0[sel], com_xy[sel], x_init[sel, 2], feet0[sel],
                               v_des[sel], w_des[sel])
        cnt[sel], swing[sel], dt[sel] = c, s, d
        X_nom[sel], X_ter[sel] = centroidal_costs(g, H, x_init[sel], v_des[sel], d)
    if len(gaits) == 1:
        g = gaits[0]
        W_X = np.tile(g.W_X, H)[None]
        W_X_ter = g.W_X_ter[None].copy()
        W_F = np.tile(g.W_F, H)[None]
    else:
        W_X = np.stack([np.tile(gaits[k].W_X, H) for k in gid])
        W_X_ter = np.stack([gaits[k].W_X_ter for k in gid])
        W_F = np.stack([np.tile(gaits[k].W_F, H) for k in gid])
    bounds = np.tile(BOUNDS_TILE, (H, 1))[None]
    return Batch(config, B, H, E, robot.mass, gaits[0].rho, cnt, dt, x_init, X_nom, X_ter,
                 W_X, W_X_ter, W_F, bounds, swing, gid, 10.0 if config == "go2_bound" else 1.0,
                 dict(seed=seed, first=first, t0=t0, v_des=v_des, gaits=[g.name for g in gaits], gait_objs=gaits, robot=robot,
                      feet0_raw=feet0_raw, w_des=w_des))


# ------------------------------------------------------------------- whole-body batches ---
SOLO12_Q0 = np.array([0, 0, 0.2409, 0, 0, 0, 1] + [0, 0.8, -1.6] * 2 + [0, -0.8, 1.6] * 2, float)  # feet at z = foot_size
FEET = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")     # abstract_cyclic_gen.py:37
HIPS = ("FL_HFE", "FR_HFE", "HL_HFE", "HR_HFE")         # abstract_cyclic_gen.py:38


@dataclass
class WholeBodyRobot:
    """What the harness constructor reads from a robot_properties config (abstract_cyclic_gen.py:28-76)"""
    name: str
    q0: np.ndarray             # nominal configuration (feet on the ground)
    feet: tuple                # eff_names
    hips: tuple                # hip_names
    mu: float = 1.0            # friction coefficient of the force projection (fista.hpp:49: 1.0 in the reference)


SOLO12_WB = WholeBodyRobot("solo12", SOLO12_Q0, FEET, HIPS)
# synthetic: robot_properties_go2/config.py:108-113,162-165 names / joint angles,
# base lowered from 0.35 so the feet rest at z = foot radius 0.02
GO2_Q0 = np.array([0, 0, 0.3168, 0, 0, 0, 1] + [0, 0.8, -1.6] * 4, float)
GO2_WB = WholeBodyRobot("go2", GO2_Q0, ("FL_foot", "FR_foot", "RL_foot", "RR_foot"),
                        ("FL_thigh_joint", "FR_thigh_joint", "RL_thigh_joint", "RR_thigh_joint"), mu=10.0)
# IK weights of the trot plan (motions/cyclic/solo12_trot.py:22-31)
TROT_IK = dict(state_wt=np.array([0., 0, 10] + [1000] * 3 + [1.0] * 12 + [0.] * 3 + [100] * 3 + [0.5] * 12),
               ctrl_wt=np.array([0, 0, 1000] + [5e2] * 3 + [1.0] * 12), swing_wt=(1e4, 1e4), cent_wt=(0.0, 5e2),
               reg_wt=(5e-2, 1e-5))


@dataclass
class WholeBodyBatch:
    """Inputs of B independent KinoDynMP.optimize(q, v, N, 1) calls as the harness prepares them."""
    dyn: Batch                 # centroidal part (x_init = [com, vcom, L] of (q, v))
    x: np.ndarray              # (B,37) [q, v]
    ik_T: int
    ik_tasks: np.ndarray       # (B, T+1, 33) kernel task blocks; com/mom refs are filled after the ADMM
    state_w: np.ndarray        # (1,36)
    ctrl_w: np.ndarray         # (1,18)
    x_reg: np.ndarray          # (B,37)
    wt_com: float
    wt_mom: float
    frame_ids: tuple

    def take(self, idx):
        """sub-batch of the problems listed in idx (copies)"""
        idx = np.asarray(idx)
        return WholeBodyBatch(self.dyn.take(idx), self.x[idx].copy(), self.ik_T, self.ik_tasks[idx].copy(), self.state_w, self.ctrl_w,
                              self.x_reg[idx].copy(), self.wt_com, self.wt_mom, self.frame_ids)


def _log3_batch(R):
    v = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    t = np.arctan2(0.5 * np.linalg.norm(v, axis=1), 0.5 * (np.trace(R, axis1=1, axis2=2) - 1.0))
    f = np.where(t < 1e-3, 0.5 * (1 + t * t / 6), t / (2 * np.sin(np.where(t < 1e-3, 1.0, t))))
    return f[:, None] * v


def make_wb_batch(model, B, first=0, seed=None, gait=TROT, ik=TROT_IK, ik_hor_ratio=0.5, wb=None):
    """Perturbed whole-body states and everything SoloMpcGaitGen.optimize hands to KinoDynMP
    (abstract_cyclic_gen.py:629-663): contact plan from the feet / CoM of (q, v), centroidal costs
    (create_costs :564-614) and the IK task list (:545-562).  Solo12 trot by default."""
    from . import fk_np
    wb = SOLO12_WB if wb is None else wb
    Q0, FEET, HIPS = wb.q0, wb.feet, wb.hips
    seed = BASE_SEED + 5 if seed is None else seed
    H = gait.horizon
    T = int(np.round(ik_hor_ratio * gait.gait_horizon * gait.gait_period / gait.gait_dt, 2))   # :128
    E = 4
    u, nrm = _draws(seed, first, B, 48)
    # state perturbation in the tangent space of q0
    dq = np.zeros((B, 18))
    dq[:, 2] = 0.01 * nrm[:, 0]
    dq[:, 3:6] = 0.05 * nrm[:, 1:4]
    dq[:, 6:] = 0.05 * nrm[:, 4:16]
    q = np.tile(Q0, (B, 1))
    q[:, 2] += dq[:, 2]
    # small rotation: quaternion of exp(w)
    w = dq[:, 3:6]
    th = np.linalg.norm(w, axis=1)
    sfac = np.where(th < 1e-8, 0.5, np.sin(0.5 * th) / np.where(th < 1e-8, 1.0, th))
    q[:, 3:6] = sfac[:, None] * w
    q[:, 6] = np.cos(0.5 * th)
    q[:, 7:] += dq[:, 6:]
    v = np.concatenate([0.1 * nrm[:, 16:19], 0.2 * nrm[:, 19:22], 0.2 * nrm[:, 22:34]], axis=1)
    t0 = np.round(0.05 * np.floor(u[:, 0] * 10), 3)
    v_des_b = np.zeros((B, 3))
    v_des_b[:, 0] = 0.3 * u[:, 1]
    kin = fk_np.kinematics(model, q, v)
    Rb = kin["oR"][0]
    v_des = np.einsum("bij,bj->bi", Rb, v_des_b)                    # :642-643
    # constructor offsets from the nominal configuration (:41-76)
    k0 = fk_np.kinematics(model, Q0[None])
    offs = np.round(fk_np.frame_positions(model, k0, HIPS)[0] - k0["com"][0], 3)
    offs[:, 1] += np.array([0.04, -0.04, 0.04, -0.04])
    yaw = np.arctan2(Rb[:, 1, 0], Rb[:, 0, 0])
    cy, sy = np.cos(yaw), np.sin(yaw)
    hip_off = np.stack([cy[:, None] * offs[None, :, 0] - sy[:, None] * offs[None, :, 1],
                        sy[:, None] * offs[None, :, 0] + cy[:, None] * offs[None, :, 1]], axis=2)   # (B,4,2)
    feet0 = np.round(fk_np.frame_positions(model, kin, FEET), 3)
    com_xy = np.round(kin["com"][:, 0:2], 3)
    robot = RobotParams(wb.name, model.total_mass, fk_np.frame_positions(model, k0, FEET)[0][:, 0:2], offs[:, 0:2],
                        float(k0["com"][0, 2]))
    cnt, swing, dt = contact_plan(gait, robot, H, t0, com_xy, kin["com"][:, 2], feet0, v_des, np.zeros(B), hip_off)
    x_init = np.concatenate([kin["com"], kin["vcom"], kin["L"]], axis=1)
    amom = _log3_batch(np.transpose(Rb, (0, 2, 1)))                   # log3(R_des R_q^T), R_des = I  (:616-627)
    X_nom, X_ter = centroidal_costs(gait, H, x_init, v_des, dt, amom)
    dyn = Batch(wb.name + "_" + gait.name + "_wb", B, H, E, model.total_mass, gait.rho, cnt, dt, x_init, X_nom, X_ter,
                np.tile(gait.W_X, H)[None], gait.W_X_ter[None].copy(), np.tile(gait.W_F, H)[None],
                np.tile(BOUNDS_TILE, (H, 1))[None], swing, np.zeros(B, dtype=np.int64), wb.mu,
                dict(seed=seed, first=first, t0=t0, v_des=v_des, v_des_body=v_des_b))
    # IK task blocks: 4 x {w, frame, ref3} | com {w, ref3} | mom {w, ref6} | state w | ctrl w
    fid = tuple(model.frame_id(n) for n in FEET)
    tasks = np.zeros((B, T + 1, 33))
    for i in range(T):
        for j in range(E):
            on = cnt[:, i, j, 0] == 1
            via = (~on) & (swing[:, i, j] == 1)
            ref = cnt[:, i, j, 1:4].copy()
            ref[via, 2] = gait.step_ht
            tasks[:, i, 5 * j] = np.where(on, ik["swing_wt"][0], np.where(via, ik["swing_wt"][1], 0.0))
            tasks[:, i, 5 * j + 1] = fid[j]
            tasks[:, i, 5 * j + 2:5 * j + 5] = ref
    tasks[:, :, 20] = ik["cent_wt"][0]
    tasks[:, :, 24] = ik["cent_wt"][1]
    tasks[:, :, 31] = ik["reg_wt"][0]
    tasks[:, :, 32] = ik["reg_wt"][1]
    x = np.concatenate([q, v], axis=1)
    x_reg = np.concatenate([np.tile(Q0, (B, 1)), np.zeros((B, 18))], axis=1)
    return WholeBodyBatch(dyn, x, T, tasks, ik["state_wt"][None].copy(), ik["ctrl_wt"][None].copy(), x_reg,
                          ik["cent_wt"][0], ik["cent_wt"][1], fid)

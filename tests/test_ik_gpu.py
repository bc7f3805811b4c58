"""GPU tests of the whole-body IK-DDP (ik_ddp.hip) through the C-ABI classes against the numpy
restatement of crocoddyl 1.9.0's SolverDDP on the reference's cost set (oracle/ik_ddp_np.py).
PARITY UNPINNED: crocoddyl / pinocchio are absent; the oracle itself is pinned by finite differences."""
import os

import numpy as np
import pytest

from bunmpc_amd import problems, urdf_model
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
ROBOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots", "solo12.json")
FEET = ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"]
Q0 = np.array([0, 0, 0.25, 0, 0, 0, 1] + [0, 0.8, -1.6] * 2 + [0, -0.8, 1.6] * 2, float)
STATE_WT = np.array([0., 0, 10] + [1000] * 3 + [1.0] * 12 + [0.] * 3 + [100] * 3 + [0.5] * 12)   # solo12_trot.py:22-23
CTRL_WT = np.array([0, 0, 1000] + [5e2] * 3 + [1.0] * 12)                                           # solo12_trot.py:25


def build_costs(ik, b, T, com_opt, mom_opt, q_reg, wt_com=0.0, wt_mom=5e2):
    """create_costs (abstract_cyclic_gen.py:545-562) + KinoDynMP::optimize's tracking tasks (kino_dyn.cpp:53-56)"""
    for i in range(T):
        for j, f in enumerate(FEET):
            if b.cnt_plan[0, i, j, 0] == 1:
                ik.add_position_tracking_task_single(f, b.cnt_plan[0, i, j, 1:4], 1e4, "cnt_0" + f, i)
            elif b.swing_time[0, i, j] == 1:
                pos = b.cnt_plan[0, i, j, 1:4].copy()
                pos[2] = problems.TROT.step_ht
                ik.add_position_tracking_task_single(f, pos, 1e4, "via_0" + f, i)
    x_reg = np.concatenate([q_reg, np.zeros(18)])
    ik.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x_reg, False)
    ik.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), False)
    ik.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x_reg, True)
    ik.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), True)
    ik.setup_costs(b.dt[0, :T])
    ik.add_centroidal_momentum_tracking_task(0, T, mom_opt[:T], wt_mom, "mom_track", False)
    ik.add_centroidal_momentum_tracking_task(0, T, mom_opt[T:T + 1], wt_mom, "mom_track_ter", True)
    ik.add_com_position_tracking_task(0, T, com_opt[:T], wt_com, "com_track", False)
    ik.add_com_position_tracking_task(0, T, com_opt[T:T + 1], wt_com, "com_track", True)


@pytest.fixture(scope="module")
def model():
    return urdf_model.RobotModel.from_json(open(ROBOT).read())


def test_ik_matches_numpy_ddp(model, oracle):
    from bunmpc_amd.inverse_kinematics_cpp import InverseKinematics
    from oracle import ik_ddp_np
    b = problems.make_batch("solo12_trot_nominal", 1)
    r = oracle.solve_batch(b, num_iters=10)
    X = r["X"][0].reshape(-1, 9)
    com_opt, mom_opt = X[:, 0:3], np.hstack([b.m * X[:, 3:6], X[:, 6:9]])
    T = 10
    x0 = np.concatenate([Q0, 0.05 * np.random.default_rng(0).standard_normal(18)])
    ref_prob = ik_ddp_np.IKProblem(model, T)
    build_costs(ref_prob, b, T, com_opt, mom_opt, Q0)
    ref = ik_ddp_np.solve_ddp(ref_prob, x0)
    ik = InverseKinematics(model, T)
    build_costs(ik, b, T, com_opt, mom_opt, Q0)
    ik.optimize(x0)
    st = ik.last_stats()
    print("gpu", st, "oracle iters", ref["iters"], ref["cost"], ref["stop"])
    assert st["status"] == 0 and ref["converged"]
    assert st["iters"] == ref["iters"]
    assert abs(st["cost"] - ref["cost"]) <= 1e-9 * abs(ref["cost"])
    xs, us = np.array(ik.get_xs()), np.array(ik.get_us())
    assert rel_l2(xs.reshape(-1), np.array(ref["xs"]).reshape(-1)) < 1e-8
    assert rel_l2(us.reshape(-1), np.array(ref["us"]).reshape(-1)) < 1e-6
    # return_opt_com / return_opt_mom against the oracle's kinematics of the same states
    from oracle import rbd_np as rb
    k = [rb.Kin(model, x[:19], x[19:]) for x in xs]
    assert np.allclose(ik.return_opt_com(), [kk.com for kk in k], atol=1e-12)
    assert np.allclose(ik.return_opt_mom(), [kk.centroidal_momentum() for kk in k], atol=1e-12)


def test_kinodyn_end_to_end(model, oracle):
    """KinoDynMP.optimize(q, v, 10, 1): centroidal ADMM from (q, v), tracking tasks, IK-DDP."""
    from bunmpc_amd.biconvex_mpc_cpp import KinoDynMP
    from oracle import ik_ddp_np, rbd_np as rb
    b = problems.make_batch("solo12_trot_nominal", 1)
    H, T = b.H, 10
    kd = KinoDynMP(model, model.total_mass, 4, H, T)
    kd.set_com_tracking_weight(np.array([0.0]))
    kd.set_mom_tracking_weight(np.array([5e2]))
    mp, ik = kd.return_dyn(), kd.return_ik()
    mp.set_rho(b.rho)
    v0 = np.zeros(18)
    kin = rb.Kin(model, Q0, v0)
    x_init = np.concatenate([kin.com, kin.vcom(), kin.centroidal_momentum()[3:]])
    for t in range(H):
        mp.set_contact_plan(b.cnt_plan[0, t], b.dt[0, t])
    mp.create_bound_constraints(b.bounds[0], 15.0, 15.0, 15.0)
    mp.create_cost_X(b.W_X[0], b.W_X_ter[0], b.X_ter[0], b.X_nom[0])
    mp.create_cost_F(b.W_F[0])
    # IK costs that the harness adds before kd.optimize (tracking tasks are added inside it)
    for i in range(T):
        for j, f in enumerate(FEET):
            if b.cnt_plan[0, i, j, 0] == 1:
                ik.add_position_tracking_task_single(f, b.cnt_plan[0, i, j, 1:4], 1e4, "cnt_0" + f, i)
    x_reg = np.concatenate([Q0, np.zeros(18)])
    ik.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x_reg, False)
    ik.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), False)
    ik.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x_reg, True)
    ik.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), True)
    ik.setup_costs(b.dt[0, :T])
    kd.compute_solve_times()
    kd.optimize(Q0, v0, 10, 1)
    # centroidal part equals the strict oracle started from the FK-derived x_init
    b.x_init[0] = x_init
    ref = oracle.solve_batch(b, num_iters=10)
    assert rel_l2(mp.return_opt_x(), ref["X"][0]) < 1e-5 and rel_l2(mp.return_opt_f(), ref["F"][0]) < 1e-5
    # IK part equals the numpy DDP on the same task set
    X = mp.return_opt_x().reshape(-1, 9)
    prob = ik_ddp_np.IKProblem(model, T)
    for i in range(T):
        for j, f in enumerate(FEET):
            if b.cnt_plan[0, i, j, 0] == 1:
                prob.add_position_tracking_task_single(f, b.cnt_plan[0, i, j, 1:4], 1e4, "cnt_0" + f, i)
    prob.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x_reg, False)
    prob.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), False)
    prob.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x_reg, True)
    prob.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), True)
    prob.setup_costs(b.dt[0, :T])
    mom = np.hstack([model.total_mass * X[:, 3:6], X[:, 6:9]])
    prob.add_centroidal_momentum_tracking_task(0, T, mom[:T], 5e2, "mom_track", False)
    prob.add_centroidal_momentum_tracking_task(0, T, mom[T:T + 1], 5e2, "mom_track_ter", True)
    prob.add_com_position_tracking_task(0, T, X[:T, 0:3], 0.0, "com_track", False)
    prob.add_com_position_tracking_task(0, T, X[T:T + 1, 0:3], 0.0, "com_track", True)
    r = ik_ddp_np.solve_ddp(prob, np.concatenate([Q0, v0]))
    xs = np.array(ik.get_xs())
    assert ik.last_stats()["iters"] == r["iters"]
    assert rel_l2(xs.reshape(-1), np.array(r["xs"]).reshape(-1)) < 1e-8
    t = kd.return_solve_times()
    assert t.shape == (3,) and t[2] >= t[0] + t[1] > 0


def _oracle_ddp(model, wb, i, X):
    """numpy DDP on problem i of a whole-body batch, tracking the centroidal solution X (H+1, 9)"""
    from oracle import ik_ddp_np
    T, names = wb.ik_T, list(model.frames)
    prob = ik_ddp_np.IKProblem(model, T)
    for t in range(T + 1):
        tk = wb.ik_tasks[i, t]
        for s in range(4):
            if tk[5 * s] != 0:
                prob._add(t, "f%d" % s, ("frame", tk[5 * s], (names[int(tk[5 * s + 1])], tk[5 * s + 2:5 * s + 5])))
        prob._add(t, "com", ("com", tk[20], X[t, 0:3]))
        prob._add(t, "mom", ("mom", tk[24], np.concatenate([wb.dyn.m * X[t, 3:6], X[t, 6:9]])))
        prob._add(t, "x", ("state", tk[31], (wb.state_w[0], wb.x_reg[i])))
        prob._add(t, "u", ("ctrl", tk[32], wb.ctrl_w[0]))
    prob.setup_costs(wb.dyn.dt[i, :T])
    return ik_ddp_np.solve_ddp(prob, wb.x[i])


def test_kinodyn_batch_matches_per_problem_oracle(model, oracle):
    """bmpc_kinodyn_solve_batch_device on perturbed whole-body states: every problem equals the
    strict centroidal oracle + numpy DDP run on that problem alone (problems finish at different
    DDP iterations inside one batch)."""
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    B = 6
    wb = problems.make_wb_batch(model, B)
    kb = KinoDynDeviceBatch(wb, model, num_iters=10)
    kb.solve()
    got = kb.results()
    ref = oracle.solve_batch(wb.dyn, num_iters=10)
    assert np.all(rel_l2(got["X"], ref["X"]) < 1e-5)
    assert np.all(got["ik_status"] == 0)
    for i in range(B):
        r = _oracle_ddp(model, wb, i, ref["X"][i].reshape(-1, 9))
        assert r["converged"] and got["ik_iters"][i] == r["iters"], (i, got["ik_iters"][i], r["iters"])
        assert abs(got["ik_cost"][i] - r["cost"]) <= 1e-7 * abs(r["cost"])
        assert rel_l2(got["xs"][i].reshape(-1), np.array(r["xs"]).reshape(-1)) < 1e-6
    print("ik iters per problem", got["ik_iters"], "loop iters", got["ddp_loop_iters"])


def test_kinodyn_batch_go2_h60(oracle):
    """BASELINE config 5 at test size: synthetic Go2 (tools/make_go2_model.py), trot, H=60, H_ik=30.
    The centroidal part is compared in the long-horizon envelope (the two CPU
    restatements themselves differ by ~1e-4 there); the IK-DDP is compared exactly, on the
    references the GPU's own centroidal solution produced."""
    import dataclasses
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    go2 = urdf_model.RobotModel.from_json(open(ROBOT.replace("solo12.json", "go2.json")).read())
    assert abs(go2.total_mass - 15.099) < 1e-9
    gait = dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0)
    B = 2
    wb = problems.make_wb_batch(go2, B, gait=gait, wb=problems.GO2_WB)
    assert wb.dyn.H == 60 and wb.ik_T == 30
    kb = KinoDynDeviceBatch(wb, go2, num_iters=10)
    kb.solve()
    got = kb.results()
    ref = oracle.solve_batch(wb.dyn, num_iters=10)
    assert np.all(rel_l2(got["X"], ref["X"]) < 5e-3) and np.all(np.isfinite(got["X"]))
    assert np.array_equal(got["stats"][:, 0], ref["stats"][:, 0])        # same number of ADMM iterations
    assert np.all(got["ik_status"] == 0)
    for i in range(B):
        r = _oracle_ddp(go2, wb, i, got["X"][i].reshape(-1, 9))
        assert r["converged"] and got["ik_iters"][i] == r["iters"], (i, got["ik_iters"][i], r["iters"])
        assert abs(got["ik_cost"][i] - r["cost"]) <= 1e-7 * abs(r["cost"])
        assert rel_l2(got["xs"][i].reshape(-1), np.array(r["xs"]).reshape(-1)) < 1e-6
    print("go2 ik iters", got["ik_iters"], "rel X", rel_l2(got["X"], ref["X"]))


@pytest.mark.parametrize("periods,ratio", [(10.0, 1.0), (16.0, 0.5)])
def test_kinodyn_batch_long_horizons(model, oracle, periods, ratio):
    """The shapes of the reference's solve-time sweep (examples/analysis/solve_times_test.py: gait horizons of 1 .. 20 periods, ik_hor_ratio
    = 1): a trot of 10 periods with the IK over the whole horizon (H = H_ik = 100) and of 16 periods (H = 160, H_ik = 80) -- the
    centroidal solve one problem per workgroup, the IK-DDP on the lock-step kernels.  Centroidal part within the long-horizon envelope
    of the CPU restatements, the IK-DDP exactly on the references the GPU's own centroidal solution produced."""
    import dataclasses
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    gait = dataclasses.replace(problems.TROT, gait_horizon=periods)
    B = 2
    wb = problems.make_wb_batch(model, B, gait=gait, ik_hor_ratio=ratio)
    assert wb.dyn.H == int(periods * 10) and wb.ik_T == int(periods * 10 * ratio)
    kb = KinoDynDeviceBatch(wb, model, num_iters=10)
    kb.solve()
    got = kb.results()
    ref = oracle.solve_batch(wb.dyn, num_iters=10)
    assert np.all(rel_l2(got["X"], ref["X"]) < 2e-2) and np.all(np.isfinite(got["X"]))
    assert np.all(got["ik_status"] == 0)
    for i in range(B):
        r = _oracle_ddp(model, wb, i, got["X"][i].reshape(-1, 9))
        assert r["converged"] and got["ik_iters"][i] == r["iters"], (i, got["ik_iters"][i], r["iters"])
        assert abs(got["ik_cost"][i] - r["cost"]) <= 1e-7 * abs(r["cost"])
        assert rel_l2(got["xs"][i].reshape(-1), np.array(r["xs"]).reshape(-1)) < 1e-6
    print("H %d H_ik %d: ik iters %s rel X %s" % (wb.dyn.H, wb.ik_T, got["ik_iters"], rel_l2(got["X"], ref["X"])))


def test_line_search_scheduling_does_not_change_results(model):
    """How the batched DDP is scheduled must not show in its results: step lengths one after the other (four problems per
    wave), four at a time (one problem per workgroup) or all ten at once (three workgroups per problem, the last to arrive
    decides) -- SolverDDP's decision, the first passing step length in its order, is the same; launches over the
    active-problem list or over all B problems touch the same problems; and the Riccati pass on one wave per problem or with
    a second wave for the gains does the same arithmetic."""
    from bunmpc_amd import _lib
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    lib = _lib.lib()
    wb = problems.make_wb_batch(model, 9)
    out = []
    old, old_all, old_gw = lib.bmpc_ik_set_speculative_below(0), lib.bmpc_ik_set_all_steps(0), lib.bmpc_ik_set_gains_wave_below(0)
    old_fd = lib.bmpc_ik_set_fused_direct_max(0)       # (a batch this small would otherwise never see the multi-kernel path)
    try:
        for below, all_steps, use_list, gains in ((0, 0, True, 0), (1 << 30, 0, True, 0), (1 << 30, 1 << 30, True, 1 << 30), (0, 0, False, 1 << 30),
                                                  (1 << 30, 1 << 30, False, 0), (6, 3, True, 4)):
            lib.bmpc_ik_set_speculative_below(below)
            lib.bmpc_ik_set_all_steps(all_steps)
            lib.bmpc_ik_set_gains_wave_below(gains)
            kb = KinoDynDeviceBatch(wb, model, num_iters=10, use_active_list=use_list)
            kb.solve()
            out.append(kb.results())
        lib.bmpc_ik_set_fused_direct_max(16)           # ... and the whole batch inside the persistent fused kernel
        kb = KinoDynDeviceBatch(wb, model, num_iters=10)
        kb.solve()
        out.append(kb.results())
        assert np.all(out[-1]["ik_fused_iters"] == out[-1]["ik_iters"]) and not np.any(out[0]["ik_fused_iters"])
    finally:
        lib.bmpc_ik_set_speculative_below(old)
        lib.bmpc_ik_set_all_steps(old_all)
        lib.bmpc_ik_set_gains_wave_below(old_gw)
        lib.bmpc_ik_set_fused_direct_max(old_fd)
    assert np.all(out[0]["ik_status"] == 0) and len(set(out[0]["ik_iters"].tolist())) > 1      # problems finish at different iterations
    for o in out[1:]:
        assert np.array_equal(out[0]["ik_iters"], o["ik_iters"])
        assert np.array_equal(out[0]["xs"], o["xs"]) and np.array_equal(out[0]["us"], o["us"])
        assert np.array_equal(out[0]["ik_cost"], o["ik_cost"]) and np.array_equal(out[0]["ik_stop"], o["ik_stop"])
        n = out[0]["ik_iters"]
        for i in range(len(n)):
            assert np.array_equal(out[0]["ik_trace"][i, :n[i]], o["ik_trace"][i, :n[i]])


def test_wide_line_search_does_not_change_results():
    """Go2 H = 60: problem 2 of the bench batch is one of the three whose line search keeps going past four step lengths (the
    CPU twin's trace: 54 of its 100 iterations).  With the active list such a problem is flagged at the first occurrence and
    from then on tries all ten step lengths at once on three workgroups; without the list it goes through the rounds of four.
    Same decisions, same bits."""
    import dataclasses
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    go2 = urdf_model.RobotModel.from_json(open(ROBOT.replace("solo12.json", "go2.json")).read())
    wb = problems.make_wb_batch(go2, 6, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    from bunmpc_amd import _lib
    lib = _lib.lib()
    out = []
    old_gw = lib.bmpc_ik_set_gains_wave_below(512)
    try:
        for use_list, gains in ((True, 512), (False, 0)):      # ... and with / without the gains wave of the Riccati pass (a pass that fails
            lib.bmpc_ik_set_gains_wave_below(gains)            # on a pivot and is started again at a larger regularisation occurs in these problems)
            kb = KinoDynDeviceBatch(wb, go2, num_iters=10, use_active_list=use_list)
            kb.solve()
            out.append(kb.results())
    finally:
        lib.bmpc_ik_set_gains_wave_below(old_gw)
    a, b = out
    assert a["ik_wide_line_search"][2] and not b["ik_wide_line_search"].any()
    tr = a["ik_trace"][2, :100, 2]
    assert a["ik_iters"][2] == 100 and ((tr > 0) & (tr < 2.0 ** -3.5)).sum() > 20       # accepted step lengths below 1/8: past the first four
    for k in ("ik_iters", "ik_status", "xs", "us", "ik_cost", "ik_stop"):
        assert np.array_equal(a[k], b[k]), k
    n = a["ik_iters"]
    for i in range(6):
        assert np.array_equal(a["ik_trace"][i, :n[i]], b["ik_trace"][i, :n[i]])


def test_riccati_pass_that_fails_and_restarts(model):
    """A negative state weight makes Q_uu indefinite: computeDirection's Cholesky meets a non-positive pivot, raises the
    regularisation and starts the pass again (solver-ddp.cpp solve()), possibly up to reg_max (status 2).  With the gains wave
    the recursion has to call it back from wherever it is; both mappings must agree bit for bit, and with the compiled twin
    on what happened."""
    import dataclasses
    from bunmpc_amd import _lib
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    from oracle import ik_oracle_c as ic
    lib = _lib.lib()
    wb = problems.make_wb_batch(model, 6)
    sw = wb.state_w.copy()
    sw[0, 24:30] = -40.0         # leg joint velocities rewarded instead of penalised
    wb = dataclasses.replace(wb, state_w=sw)
    out = []
    old_gw, old_fd = lib.bmpc_ik_set_gains_wave_below(0), lib.bmpc_ik_set_fused_direct_max(0)
    try:
        for gains in (0, 1 << 30):
            lib.bmpc_ik_set_gains_wave_below(gains)
            kb = KinoDynDeviceBatch(wb, model, num_iters=10)
            kb.solve()
            out.append(kb.results())
        lib.bmpc_ik_set_fused_direct_max(16)           # the same restarts inside the fused kernel's tick loop
        kb = KinoDynDeviceBatch(wb, model, num_iters=10)
        kb.solve()
        fused = kb.results()
    finally:
        lib.bmpc_ik_set_gains_wave_below(old_gw)
        lib.bmpc_ik_set_fused_direct_max(old_fd)
    a, b = out
    for k in ("ik_iters", "ik_status", "xs", "us", "ik_cost", "ik_stop"):
        assert np.array_equal(fused[k], a[k]), k
    n = a["ik_iters"]
    reg = [a["ik_trace"][i, :n[i], 1] for i in range(6)]
    assert any((r[1:] > 5 * r[:-1]).any() for r in reg if len(r) > 1) or (a["ik_status"] == 2).any()     # the regularisation did go up
    for k in ("ik_iters", "ik_status", "xs", "us", "ik_cost", "ik_stop"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    for i in range(6):
        assert np.array_equal(a["ik_trace"][i, :n[i]], b["ik_trace"][i, :n[i]], equal_nan=True)
    # the problem is unbounded below (the cost runs off to -1e7) and the two implementations part ways by rounding after ~30
    # iterations; up to there they must have taken the same decisions: regularisation (incl. the failed passes' increases),
    # accepted step lengths, costs
    tw = ic.solve_wb_batch(ic.Model(model), wb, a["X"], trace=True)
    K = 18
    for i in range(6):
        g, t = a["ik_trace"][i, :K], tw["trace"][i, :K]
        assert n[i] >= K and tw["iters"][i] >= K
        assert np.array_equal(g[:, 1], t[:, 1]) and np.array_equal(g[:, 2], t[:, 2]), (i, g[:, 1:3], t[:, 1:3])
        assert np.all(np.abs(g[:, 0] - t[:, 0]) <= 1e-6 * np.abs(t[:, 0]))
        assert (g[:, 1] > 1e-2).any()        # within the compared prefix the regularisation has already gone up


@pytest.mark.parametrize("T", [63, 64, 130, 200])
def test_ik_long_horizons(model, T):
    """n_col = 63 (T + 1 = 64 nodes: the last horizon the fused single-launch kernel takes), 64, 130 and 200 (the horizons of the
    reference's solve_times_test.py with ik_hor_ratio = 1: the lock-step kernels, node costs and gaps 64 nodes at a time):
    regularisation + one foot target every fourth node, the numpy DDP on the same problem as the reference"""
    from bunmpc_amd.inverse_kinematics_cpp import InverseKinematics
    from oracle import ik_ddp_np
    x0 = np.concatenate([Q0, np.zeros(18)])
    target = np.array([0.21, 0.15, 0.03])

    def costs(ik):
        for i in range(0, T, 4):
            ik.add_position_tracking_task_single("FL_FOOT", target, 1e3, "t%d" % i, i)
        ik.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x0, False)
        ik.add_ctrl_regularization_cost(0, T, 1e-5, "uReg", CTRL_WT, np.zeros(18), False)
        ik.add_state_regularization_cost(0, T, 5e-2, "xReg", STATE_WT, x0, True)
        ik.setup_costs(np.full(T, 0.02))

    ik = InverseKinematics(model, T)
    costs(ik)
    ik.optimize(x0)
    st = ik.last_stats()
    ref_prob = ik_ddp_np.IKProblem(model, T)
    costs(ref_prob)
    ref = ik_ddp_np.solve_ddp(ref_prob, x0)
    assert st["status"] == 0 and st["iters"] == ref["iters"], (st, ref["iters"])
    assert abs(st["cost"] - ref["cost"]) <= 1e-9 * abs(ref["cost"])
    assert rel_l2(np.array(ik.get_xs()).reshape(-1), np.array(ref["xs"]).reshape(-1)) < 1e-8
    with pytest.raises(Exception):
        InverseKinematics(model, 256)


IK_GOLDEN = sorted(__import__("glob").glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ik_*.npz")))


@pytest.fixture(params=["fused", "lockstep"])
def small_batch_path(request):
    """batches of <= 16 problems run entirely inside the fused kernel by default (bmpc_ik_set_fused_direct_max); the multi-kernel
    path must give the same answers on them"""
    from bunmpc_amd import _lib
    old = _lib.lib().bmpc_ik_set_fused_direct_max(16 if request.param == "fused" else 0)
    yield request.param
    _lib.lib().bmpc_ik_set_fused_direct_max(old)


@pytest.mark.parametrize("path", IK_GOLDEN, ids=[os.path.basename(p)[:-4] for p in IK_GOLDEN])
def test_ik_golden_fixtures(path, small_batch_path):
    """Committed inputs / outputs of the whole-body DDP (tests/golden/make_golden_ik.py: four Solo12 problems, three synthetic-Go2
    H = 60 / H_ik = 30 problems of which one runs to SolverDDP's maxiter), fed to bmpc_ik_solve_batch_device as the arrays the
    file holds -- no oracle build, no problem generator between the fixture and the kernels.  The GPU must take the committed
    discrete path (iteration count, status, every accepted step length and regularisation value) to the committed
    trajectories.  PARITY UNPINNED (the fixtures are the CPU twin's outputs, not the reference's)."""
    from tests.golden import make_golden_ik as mg
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    import torch
    g = np.load(path)
    B = g["x0"].shape[0]
    model, wb = mg.wb_batch(str(g["robot"]), B)
    assert wb.ik_T == int(g["T"]) and wb.dyn.H == int(g["H"])
    kb = KinoDynDeviceBatch(wb, model, num_iters=10)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(kb.device)      # noqa: E731
    for dst, k in ((kb.x, "x0"), (kb.dt_ik, "dt"), (kb.tasks, "tasks"), (kb.state_w, "state_w"), (kb.x_reg, "x_reg"), (kb.ctrl_w, "ctrl_w")):
        assert tuple(dst.shape) == g[k].shape, k
        dst.copy_(up(g[k]))
    kb.solve_ik_only()
    r = kb.results()
    assert np.array_equal(r["ik_iters"], g["iters"]) and np.array_equal(r["ik_status"], g["status"])
    for i in range(B):
        n = int(g["iters"][i])
        assert np.array_equal(r["ik_trace"][i, :n, 1:3], g["trace"][i, :n, 1:3]), i
        assert np.all(np.abs(r["ik_trace"][i, :n, 0] - g["trace"][i, :n, 0]) <= 1e-6 * np.abs(g["trace"][i, :n, 0])), i
    assert np.all(np.abs(r["ik_cost"] - g["cost"]) <= 1e-8 * np.abs(g["cost"]))
    e = rel_l2(r["xs"].reshape(B, -1), g["xs"].reshape(B, -1))
    eu = rel_l2(r["us"].reshape(B, -1), g["us"].reshape(B, -1))
    print("%s: xs rel-L2 vs fixture max %.2e, us max %.2e, iterations %s" % (os.path.basename(path), e.max(), eu.max(), r["ik_iters"].tolist()))
    assert np.all(e < 1e-6) and np.all(eu < 1e-4)


def test_list_index_checks_report_instead_of_faulting(model):
    """An out-of-range entry in the active-problem list (injected through bmpc_ik_sched_t.debug_inject) is never used as an
    index: the solve returns BMPC_DEVICE_ERROR naming the check, the process lives, and the next solve of the same objects is
    clean.  (Round 2 lost a test run to an abort inside this code while it was being written; the checks are what would
    have turned that into an error message.)"""
    from bunmpc_amd import _lib
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    wb = problems.make_wb_batch(model, 8)
    kb = KinoDynDeviceBatch(wb, model, num_iters=10)
    kb.solve()
    good = kb.results()
    kb.set_schedule(debug_inject=1)
    with pytest.raises(_lib.BmpcError) as e:
        kb.solve()
    assert e.value.code == _lib.DEVICE_ERROR and "index check" in str(e.value) and "entry out of range" in str(e.value)
    kb.set_schedule(debug_inject=0)
    kb.solve()
    again = kb.results()
    assert np.array_equal(again["xs"], good["xs"]) and np.array_equal(again["ik_iters"], good["ik_iters"])


def test_express_lane_does_not_change_results(model):
    """The express lane (ik_select_kernel + the persistent fused kernel on a side stream): on the Solo12 batch it takes the problems
    farthest from converging off the active list after three iterations and runs them to the end in one launch.  Same device
    code as the multi-kernel path, so every bit of every problem's result must be the same with the lane and without it --
    for the problems the lane took and for those it left."""
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    wb = problems.make_wb_batch(model, 1024)
    out = {}
    for name, cap in (("off", -1), ("on", 96)):
        kb = KinoDynDeviceBatch(wb, model, num_iters=10, schedule={"express_cap": cap})
        kb.solve()
        out[name] = kb.results()
        kb.solve()                                   # a second solve of the same objects: the lane's device state starts afresh
        again = kb.results()
        assert np.array_equal(again["xs"], out[name]["xs"]) and np.array_equal(again["ik_fused_iters"], out[name]["ik_fused_iters"])
    off, on = out["off"], out["on"]
    took = on["ik_fused_iters"] > 0
    print("express lane: took %d of 1024 problems (DDP iterations of those: %s...), host loop %d -> %d iterations"
          % (took.sum(), sorted(on["ik_iters"][took].tolist())[-5:], off["ddp_loop_iters"], on["ddp_loop_iters"]))
    assert not np.any(off["ik_fused_iters"]) and 16 <= took.sum() <= 96
    longest = np.argsort(-on["ik_iters"])[:8]
    assert np.all(took[longest])                     # what it is for: the longest-running problems are on the lane
    assert on["ddp_loop_iters"] < off["ddp_loop_iters"]
    for k in ("xs", "us", "ik_cost", "ik_stop", "ik_iters", "ik_status", "X", "F"):
        assert np.array_equal(on[k], off[k]), k
    n = on["ik_iters"]
    for i in np.where(took)[0][:6]:                  # ... the per-iteration trace too: same accepted step lengths, same regularisation
        assert np.array_equal(on["ik_trace"][i, :n[i]], off["ik_trace"][i, :n[i]]), i
    assert np.all(on["ik_fused_iters"][took] <= n[took]) and np.all(on["ik_fused_iters"][took] >= n[took] - 6)


def test_fused_kernel_on_the_long_horizon_and_through_regularisation_restarts():
    """The persistent fused kernel where the express lane's own rule never sends it: forced onto the synthetic Go2 H = 60 /
    H_ik = 30 batch from iteration 2 (31 nodes: the producer waves run sixteen pairs ahead of the recursion, problems take
    partial steps, second rounds of step lengths and up to SolverDDP's 100 iterations) -- and, on Solo12, onto problems whose
    first Riccati pass fails on a pivot and restarts at a larger regularisation inside the kernel.  Every bit of every result as
    without the lane."""
    import dataclasses
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    robots = os.path.dirname(ROBOT)
    go2 = urdf_model.RobotModel.from_json(open(os.path.join(robots, "go2.json")).read())
    wb = problems.make_wb_batch(go2, 192, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    out = {}
    for name, sched in (("off", {"express_cap": -1}), ("forced", {"express_cap": 24, "debug_inject": 2})):
        kb = KinoDynDeviceBatch(wb, go2, num_iters=10, schedule=sched)
        kb.solve()
        out[name] = kb.results()
    off, on = out["off"], out["forced"]
    took = on["ik_fused_iters"] > 0
    print("Go2 H_ik=30: the forced lane took %d problems, iterations %s" % (took.sum(), sorted(on["ik_iters"][took].tolist())))
    assert took.sum() == 24 and on["ik_iters"][took].max() >= 60
    for k in ("xs", "us", "ik_cost", "ik_stop", "ik_iters", "ik_status"):
        assert np.array_equal(on[k], off[k]), k
    n = on["ik_iters"]
    for i in np.where(took)[0]:
        assert np.array_equal(on["ik_trace"][i, :min(n[i], 128)], off["ik_trace"][i, :min(n[i], 128)]), i
    assert np.any(on["ik_trace"][took][:, :, 2][on["ik_trace"][took][:, :, 2] > 0] < 1.0)      # partial steps were taken inside the kernel


def test_fused_kernel_restarts_its_riccati_pass(model):
    """heavy state-regularisation-free problems whose first backward pass hits a non-positive pivot: the restart (regularisation
    x 10, same derivatives) happens inside the fused kernel's tick loop"""
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    wb = problems.make_wb_batch(model, 128)
    wb.ik_tasks[:, :, 31] = 0.0          # no state regularisation: Q_uu of the first pass is not positive definite
    out = {}
    for name, sched in (("off", {"express_cap": -1}), ("forced", {"express_cap": 16, "debug_inject": 2})):
        kb = KinoDynDeviceBatch(wb, model, num_iters=10, ddp_maxiter=30, schedule=sched)
        kb.solve()
        out[name] = kb.results()
    off, on = out["off"], out["forced"]
    took = on["ik_fused_iters"] > 0
    reg = off["ik_trace"][:, :, 1]
    print("forced lane took %d problems; largest regularisation along the way %.1e; statuses %s" % (took.sum(), reg.max(), np.unique(on["ik_status"]).tolist()))
    assert took.sum() == 16
    for k in ("xs", "us", "ik_cost", "ik_iters", "ik_status"):
        assert np.array_equal(on[k], off[k]), k
    n = on["ik_iters"]
    for i in np.where(took)[0]:
        assert np.array_equal(on["ik_trace"][i, :n[i]], off["ik_trace"][i, :n[i]]), i


def test_mappings_agree_bit_for_bit_at_full_size(model):
    """4096 problems through every scheduling of the DDP -- line search always four problems per wave, always four step lengths
    per problem, the default thresholds, and the default with the express lane -- must give the same bits in every output,
    the reported cost included.  (With hipcc's default -ffp-contract=fast they did not quite: the back end fused multiplies into
    adds of neighbouring statements differently in the different instantiations of the same source, and 3-14 problems of 4096
    came out with a final cost one ulp apart; ik_ddp.hip is compiled with -ffp-contract=on, bunmpc_amd/build.py.)"""
    from bunmpc_amd import _lib
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    lib = _lib.lib()
    wb = problems.make_wb_batch(model, 4096, seed=4196)
    out = {}
    for name, below, cap in (("default", 1024, -1), ("never_spec", 0, -1), ("always_spec", 1 << 30, -1), ("express", 1024, 96)):
        old = lib.bmpc_ik_set_speculative_below(below)
        try:
            kb = KinoDynDeviceBatch(wb, model, num_iters=10, schedule={"express_cap": cap})
            kb.solve()
            out[name] = kb.results()
        finally:
            lib.bmpc_ik_set_speculative_below(old)
    assert (out["express"]["ik_fused_iters"] > 0).sum() == 96
    for name in ("never_spec", "always_spec", "express"):
        for k in ("xs", "us", "ik_cost", "ik_stop", "ik_iters", "ik_status"):
            d = np.where(np.any((out["default"][k] != out[name][k]).reshape(4096, -1), axis=1))[0]
            assert len(d) == 0, (name, k, d[:8].tolist())

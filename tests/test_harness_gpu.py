"""GPU tests of the harness mirror (bunmpc_amd/cyclic_gen.py :: SoloMpcGaitGen): one MPC call through
the drop-in classes equals the same problem solved as a row of the batch entry point, and the plan it
returns has the reference's shape and end-point conventions (abstract_cyclic_gen.py:629-698)."""
import os
import types

import numpy as np
import pytest

from bunmpc_amd import problems, urdf_model
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
ROBOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots", "solo12.json")


def trot_params():
    """motions/cyclic/solo12_trot.py:12-41 as a BiconvexMotionParams-shaped object"""
    g, ik = problems.TROT, problems.TROT_IK
    return types.SimpleNamespace(
        gait_period=g.gait_period, stance_percent=list(g.stance_percent), gait_dt=g.gait_dt,
        phase_offset=list(g.phase_offset), step_ht=g.step_ht, nom_ht=g.nom_ht, gait_horizon=g.gait_horizon,
        W_X=g.W_X, W_X_ter=g.W_X_ter, W_F=g.W_F, rho=g.rho, ori_correction=list(g.ori_correction),
        swing_wt=list(ik["swing_wt"]), cent_wt=list(ik["cent_wt"]), reg_wt=list(ik["reg_wt"]),
        state_wt=ik["state_wt"], ctrl_wt=list(ik["ctrl_wt"]))


def test_one_mpc_call_equals_the_batch_row():
    from bunmpc_amd.cyclic_gen import SoloMpcGaitGen
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    model = urdf_model.RobotModel.from_json(open(ROBOT).read())
    B = 3
    wb = problems.make_wb_batch(model, B)
    kb = KinoDynDeviceBatch(wb, model, num_iters=10)
    kb.solve()
    got = kb.results()
    x_reg = np.concatenate([problems.SOLO12_Q0, np.zeros(18)])
    gg = SoloMpcGaitGen(model, model, x_reg, 0.05, problems.SOLO12_Q0)
    for i in range(B):
        gg.update_gait_params(trot_params(), wb.dyn.meta["t0"][i])
        q, v = wb.x[i, :19].copy(), wb.x[i, 19:].copy()
        q[0:2] = 0
        v_des_body = wb.dyn.meta["v_des_body"][i]
        xs_int, us_int, f_int = gg.optimize(q, v, wb.dyn.meta["t0"][i], v_des_body, 0.0, dyn_iters=10)
        # the harness's own inputs are the batch row
        assert np.allclose(gg.cnt_plan, wb.dyn.cnt_plan[i], rtol=0, atol=1e-15) and np.array_equal(gg.dt_arr, wb.dyn.dt[i])
        assert np.array_equal(gg.swing_time, wb.dyn.swing_time[i])
        assert np.allclose(gg.X_nom, wb.dyn.X_nom[i], rtol=0, atol=1e-15) and np.allclose(gg.X_ter, wb.dyn.X_ter[i], rtol=0, atol=1e-15)
        # ... and so are the solutions
        assert rel_l2(gg.mp.return_opt_x(), got["X"][i]) < 1e-9
        assert rel_l2(np.array(gg.ik.get_xs()).reshape(-1), got["xs"][i].reshape(-1)) < 1e-7
        assert gg.ik.last_stats()["iters"] == got["ik_iters"][i]
        # 1 kHz plan: size = min(10, int(0.05/0.05) + 2) = 3 intervals of 50 rows (t0 is a knot time here)
        assert gg.size == 3 and xs_int.shape == (150, 37) and us_int.shape == (150, 18) and f_int.shape == (150, 12)
        xs = np.array(gg.ik.get_xs())
        assert np.array_equal(xs_int[0], xs[0]) and np.array_equal(xs_int[49], xs[1]) and np.array_equal(xs_int[50], xs[1])
        F = gg.mp.return_opt_f().reshape(-1, 12)
        assert np.array_equal(f_int[0], F[0]) and np.array_equal(f_int[149], F[3])


def test_turning_uses_the_composite_inertia():
    """w_des != 0: yaw momentum reference = (I_composite_b [0,0,w])[2] in X_nom and X_ter (:603-608)"""
    from bunmpc_amd.cyclic_gen import SoloMpcGaitGen, composite_inertia_base
    model = urdf_model.RobotModel.from_json(open(ROBOT).read())
    x_reg = np.concatenate([problems.SOLO12_Q0, np.zeros(18)])
    gg = SoloMpcGaitGen(model, model, x_reg, 0.05, problems.SOLO12_Q0)
    gg.update_gait_params(trot_params(), 0.0)
    q, v = problems.SOLO12_Q0.copy(), np.zeros(18)
    xs_int, _, f_int = gg.optimize(q, v, 0.0, np.array([0.2, 0.0, 0.0]), 0.5, dyn_iters=10)
    Izz = composite_inertia_base(model, problems.SOLO12_Q0)[2, 2]
    assert np.allclose(gg.X_nom[8::9], 0.5 * Izz) and gg.X_ter[8] == gg.X_nom[8]
    assert np.all(np.isfinite(xs_int)) and np.all(np.isfinite(f_int))
    # stance feet of a trot at t=0 (FL, HR; phase offsets 0) carry the weight
    F0 = gg.mp.return_opt_f().reshape(-1, 4, 3)[0]
    assert F0[:, 2].sum() > 0.5 * model.total_mass * 9.81


def test_batched_mpc_equals_the_harness_call_by_call():
    """BatchedMpc.optimize (states -> 1 kHz plans, all on the device) against SoloMpcGaitGen.optimize per robot"""
    from bunmpc_amd.cyclic_gen import SoloMpcGaitGen
    from bunmpc_amd.mpc_batch import BatchedMpc
    model = urdf_model.RobotModel.from_json(open(ROBOT).read())
    B = 5
    wb = problems.make_wb_batch(model, B)
    t0, vb = wb.dyn.meta["t0"], wb.dyn.meta["v_des_body"]
    out = BatchedMpc(model, dyn_iters=10).optimize(wb.x, t0, vb)
    x_reg = np.concatenate([problems.SOLO12_Q0, np.zeros(18)])
    gg = SoloMpcGaitGen(model, model, x_reg, 0.05, problems.SOLO12_Q0)
    rows = out["rows"].cpu().numpy()
    for i in range(B):
        gg.update_gait_params(trot_params(), t0[i])
        q, v = wb.x[i, :19].copy(), wb.x[i, 19:].copy()
        xs_int, us_int, f_int = gg.optimize(q, v, t0[i], vb[i], 0.0, dyn_iters=10)
        assert rows[i] == xs_int.shape[0]
        for name, ref in (("xs_int", xs_int), ("us_int", us_int), ("f_int", f_int)):
            got = out[name][i, :rows[i]].cpu().numpy()
            assert rel_l2(got.reshape(-1), ref.reshape(-1)) < 1e-7, (i, name)


def test_batched_mpc_reuses_its_buffers_without_changing_results():
    """a second optimize() of the same size runs in the first call's plan tensors / solver workspace (nothing allocated):
    same result, bit for bit, as a fresh object; device tensors are accepted as inputs"""
    import torch
    from bunmpc_amd.mpc_batch import BatchedMpc
    model = urdf_model.RobotModel.from_json(open(ROBOT).read())
    wb1, wb2 = problems.make_wb_batch(model, 6, seed=11), problems.make_wb_batch(model, 6, seed=12)
    mpc = BatchedMpc(model, dyn_iters=10)
    first = mpc.optimize(wb1.x, wb1.dyn.meta["t0"], wb1.dyn.meta["v_des_body"])
    keep = first["xs_int"].clone()
    dev = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda:0")
    again = mpc.optimize(dev(wb2.x), dev(wb2.dyn.meta["t0"]), dev(wb2.dyn.meta["v_des_body"]))
    fresh = BatchedMpc(model, dyn_iters=10).optimize(wb2.x, wb2.dyn.meta["t0"], wb2.dyn.meta["v_des_body"])
    for k in ("xs_int", "us_int", "f_int", "rows", "xs", "us", "X", "F"):
        assert torch.equal(again[k], fresh[k]), k
    assert not torch.equal(keep, again["xs_int"])
    assert again["xs"].data_ptr() == first["xs"].data_ptr()      # the workspace was reused


def crouch_plan(model):
    """An ACyclicMotionParams-shaped plan (weight_abstract.py:44-80): stand, then crouch -- two phases of nominal height,
    regularisation posture and weights, so the per-node regularisation vectors differ along the horizon."""
    q_stand = problems.SOLO12_Q0.copy()
    q_crouch = q_stand.copy()
    q_crouch[2] -= 0.05
    q_crouch[7:] = np.array([0, 1.0, -2.0] * 2 + [0, -1.0, 2.0] * 2)
    feet = np.array([[0.1946, 0.14695, 0.018], [0.1946, -0.14695, 0.018], [-0.1946, 0.14695, 0.018], [-0.1946, -0.14695, 0.018]])
    p = types.SimpleNamespace()
    p.n_col, p.dt_arr, p.plan_freq = 12, np.full(12, 0.05), [[0.3, 0, 2.0]]
    p.cnt_plan = [[[1.0, *feet[j], 0.0, 2.0] for j in range(4)]]
    p.W_X = np.array([1e-5, 1e-5, 1e5, 1e1, 1e1, 2e2, 1e4, 1e4, 1e4])
    p.W_X_ter = 10 * p.W_X
    p.W_F = np.array(4 * [1e1, 1e1, 1e1])
    p.rho = 5e4
    p.X_nom = [[0, 0, 0.22, 0, 0, 0, 0, 0, 0, 0.0, 0.3], [0, 0, 0.17, 0, 0, 0, 0, 0, 0, 0.3, 2.0]]
    p.X_ter = np.array([0, 0, 0.17, 0, 0, 0, 0, 0, 0.0])
    p.bounds = [[-0.25, -0.25, 0.1, 0.25, 0.25, 0.3, 0.0, 2.0]]
    p.cnt_wt, p.swing_wt, p.cent_wt = 1e3, None, [1e-1, 5e1]
    sw1 = np.array([0., 0, 10] + [1000] * 3 + [1.0] * 12 + [0.] * 3 + [100] * 3 + [0.5] * 12)
    sw2 = np.array([0., 0, 100] + [500] * 3 + [10.0] * 12 + [0.] * 3 + [100] * 3 + [1.0] * 12)
    p.state_reg = [np.hstack([q_stand, np.zeros(18), 0.0, 0.3]), np.hstack([q_crouch, np.zeros(18), 0.3, 2.0])]
    p.state_wt = [np.hstack([sw1, 0.0, 0.3]), np.hstack([sw2, 0.3, 2.0])]
    p.state_scale = [[1e-2, 0.0, 0.3], [5e-2, 0.3, 2.0]]
    cw = np.array([1e-2, 0, 1000] + [5e2] * 3 + [1.0] * 12)
    p.ctrl_wt = [np.hstack([cw, 0.0, 0.3]), np.hstack([2 * cw, 0.3, 2.0])]
    p.ctrl_reg = [np.hstack([np.zeros(18), 0.0, 0.3]), np.hstack([np.zeros(18), 0.3, 2.0])]
    p.ctrl_scale = [[1e-4, 0.0, 0.3], [1e-4, 0.3, 2.0]]
    p.kp, p.kd = [[2.5, 0.0, 2.0]], [[0.08, 0.0, 2.0]]
    return p, q_stand


def test_acyclic_generator_with_time_varying_regularisation():
    """SoloAcyclicGen (abstract_acyclic_gen.py) on a stand -> crouch plan: the per-node regularisation vectors reach the
    kernels, and the IK solution equals the numpy DDP given the same task list and the GPU's centroidal solution"""
    from bunmpc_amd.acyclic_gen import SoloAcyclicGen
    from oracle import ik_ddp_np
    model = urdf_model.RobotModel.from_json(open(ROBOT).read())
    plan, q0 = crouch_plan(model)
    v0 = np.zeros(18)
    gen = SoloAcyclicGen(model, model)
    gen.dyn_iters = 10
    gen.update_motion_params(plan, q0, 0.0)
    xs_int, us_int, f_int = gen.optimize(q0.copy(), v0, 0.1)
    T = plan.n_col
    assert xs_int.shape == (T * 50, 37) and us_int.shape == (T * 50, 18) and f_int.shape == (T * 50, 12)
    xs = np.array(gen.ik.get_xs())
    assert np.array_equal(xs_int[0], xs[0]) and np.array_equal(xs_int[49], xs[0]) and np.array_equal(xs_int[50], xs[1])   # zero-order hold
    assert gen.get_plan_freq(0.5) == 0.3 and gen.get_gains(0.5) == (2.5, 0.08)

    class FakeKd:   # same cost calls, recorded by the numpy problem instead of the GPU handle
        def __init__(self):
            self.ikp = ik_ddp_np.IKProblem(model, T)
            self.dyn = types.SimpleNamespace(**{k: (lambda *a, **kw: None) for k in
                                                ("set_rho", "set_contact_plan", "create_bound_constraints", "create_cost_X", "create_cost_F")})
        def set_com_tracking_weight(self, w): pass
        def set_mom_tracking_weight(self, w): pass
        def return_ik(self): return self.ikp
        def return_dyn(self): return self.dyn

    ref_gen = SoloAcyclicGen(model, model)
    fake = FakeKd()
    ref_gen._make_kd = lambda: fake
    ref_gen.ee_frame_id = list(ref_gen.eff_names)          # the numpy problem addresses frames by name
    ref_gen.update_motion_params(plan, q0, 0.0)
    ref_gen.create_contact_plan(q0.copy(), v0, 0.1)
    ref_gen.create_costs(q0.copy(), v0, 0.1)
    assert np.array_equal(ref_gen.cnt_plan, gen.cnt_plan) and np.array_equal(ref_gen.X_nom, gen.X_nom)
    X = gen.mp.return_opt_x().reshape(-1, 9)
    mom = np.hstack([model.total_mass * X[:, 3:6], X[:, 6:9]])
    prob = fake.ikp                                           # tracking tasks as KinoDynMP::optimize adds them (kino_dyn.cpp:50-56)
    prob.add_centroidal_momentum_tracking_task(0, T, mom[:T], plan.cent_wt[1], "mom_track", False)
    prob.add_centroidal_momentum_tracking_task(0, T, mom[T:T + 1], plan.cent_wt[1], "mom_track_ter", True)
    prob.add_com_position_tracking_task(0, T, X[:T, 0:3], plan.cent_wt[0], "com_track", False)
    prob.add_com_position_tracking_task(0, T, X[T:T + 1, 0:3], plan.cent_wt[0], "com_track", True)
    r = ik_ddp_np.solve_ddp(prob, np.concatenate([q0, v0]))
    assert gen.ik.last_stats()["iters"] == r["iters"] and r["converged"]
    assert rel_l2(xs.reshape(-1), np.array(r["xs"]).reshape(-1)) < 1e-8
    # the other line-search mapping (four problems per wave, one wave) reads the per-node vectors the same way
    from bunmpc_amd import _lib
    old = _lib.lib().bmpc_ik_set_speculative_below(0)
    try:
        gen.optimize(q0.copy(), v0, 0.1)
    finally:
        _lib.lib().bmpc_ik_set_speculative_below(old)
    assert np.array_equal(np.array(gen.ik.get_xs()), xs)
    # the regularisation really differs along the horizon: the late nodes sit near the crouch posture
    assert abs(xs[-1][8] - 1.0) < abs(xs[-1][8] - 0.8)

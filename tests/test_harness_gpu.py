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

"""Parity checks the round-1 review asked for (VERDICT.md "Next round" item 1), all through the C-ABI on the GPU:
 (a) the fp32 kernel (BASELINE config 3) against the CPU ORACLE, not against its fp64 sibling;
 (c) whole-body DDP problems that run into crocoddyl's maxiter = 100 (synthetic Go2, H = 60 / H_ik = 30): the GPU and the
     compiled CPU twin must take the same 100 iterations -- same accepted step lengths, same regularisation sequence, same cost;
 (d) full-size property checks for the Go2 shapes (H = 40, B = 4096, fp64 and fp32; H = 60 / H_ik = 30, B = 1024 with the IK).
The measured-spread envelope of item (b) lives in tests/util.py and is used by tests/test_biconvex_gpu.py.
PARITY UNPINNED throughout: the reference holds no vectors for this path and cannot be built here."""
import dataclasses
import os

import numpy as np
import pytest

from bunmpc_amd import batch as bb
from bunmpc_amd import problems, urdf_model
from tests.util import chaos_ensemble, cpu_spread, prefix_parity, rel_l2, within_envelope

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROBOTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots")

# fp32 tolerance against the fp64 CPU oracle.  The iterates are fp32 (unit round-off 6e-8) through ~2 500 FISTA iterations per
# solve; every decision is taken in fp64.  Measured on the MI355X (tools/fp32_diag.py, 1024 problems per shape): median 2e-6;
# synthetic Go2 bound H = 40: every problem below 4e-5; Solo12 trot: 98.7 % of the problems below 1e-5 -- inside north_star's
# fp64 tolerance -- and 1.3 % between 1e-4 and 5e-3.  Those stop their force FISTA a few iterations away from the oracle:
# its exit test ||y+ - y|| < 1e-5 sits at the resolution of an fp32 force iterate (240 components of ~10 N x 6e-8), a property
# of running the reference's absolute tolerance in fp32, not of the kernel (the ADMM count stays the oracle's).  Held to:
# median <= 5e-6, 95 % <= 1e-5, every calm problem <= 1e-2; problems in the chaotic regime of the reference algorithm
# (tests/util.py), where a 1e-7 perturbation is amplified like any other, must keep the oracle's ADMM count, stay finite and
# satisfy the invariants.  (The first version of this test caught a real defect: the fp32 backtracking test took A d as the
# difference of two rounded images and retried for ever on ~1.5 % of the trot problems -- rel. error 0.5; biconvex_admm.hip.)
FP32_MEDIAN, FP32_P95, FP32_MAX = 5e-6, 1e-5, 1e-2


@pytest.mark.parametrize("config,B,H", [("go2_bound", 256, 40), ("solo12_trot", 256, None), ("solo12_mixed", 96, None)])
def test_fp32_kernel_against_the_cpu_oracle(oracle, config, B, H):
    b = problems.make_batch(config, B, H=H) if H else problems.make_batch(config, B)
    ref, spread = cpu_spread(b, 10, oracle, with_numpy=False)
    got = bb.solve_host(b, num_iters=10, precision="f32")
    assert np.array_equal(got["stats"][:, [0, 5]], ref["stats"][:, [0, 5]])          # the oracle's ADMM count, nothing diverged
    err = np.maximum(rel_l2(got["X"], ref["X"]), rel_l2(got["F"], ref["F"]))
    calm = spread <= 1e-9
    print("%s fp32 vs CPU oracle: calm %d problems median %.2e p95 %.2e max %.2e | chaotic %d problems max %.2e (CPU spread max %.2e)"
          % (config, calm.sum(), np.median(err[calm]), np.quantile(err[calm], 0.95), err[calm].max(), (~calm).sum(),
             err[~calm].max() if (~calm).any() else 0.0, spread.max()))
    assert np.median(err[calm]) <= FP32_MEDIAN and np.quantile(err[calm], 0.95) <= FP32_P95 and np.all(err[calm] <= FP32_MAX)
    if config != "solo12_mixed":
        assert calm.mean() > 0.9
    assert np.all(np.isfinite(got["X"])) and np.all(np.isfinite(got["F"]))
    F = got["F"].reshape(B, b.H, b.E, 3)
    assert np.all(F[b.cnt_plan[..., 0] == 0] == 0.0) and np.all(F[..., 2] >= 0)
    # the fp64 residual check of config 3: the violation the kernel reports, re-derived in fp64 from the returned iterates
    for i in range(0, B, 16):
        A, bf = oracle.dense_A_f(b.cnt_plan[i], b.dt[i], b.m, got["F"][i], b.x_init[i])
        r = np.linalg.norm(A @ got["X"][i] - bf)
        assert abs(r - got["dyn_viol"][i]) <= 1e-4 * max(r, 1e-3), (i, r, got["dyn_viol"][i])


def _sampled_prefix_parity(name, b, sub, iters, got, oracle, label, calm_tol=1e-12):
    """The GPU's sampled problems against the strict oracle under the per-problem CPU ensembles of tests/golden/chaos_<name>.npz
    (tools/chaos_ensemble.py; tests/util.py::prefix_parity): exact discrete path and violation to 1e-9 before the reference
    algorithm's own bifurcation, the problem's OWN ensemble range after it.  The ensemble is recomputed here from the C members
    and must reproduce the committed file (the numpy twin's contribution comes from the file)."""
    g = np.load(os.path.join(GOLDEN, "chaos_%s.npz" % name))
    assert np.array_equal(g["sub"], sub) and int(g["iters"]) == iters
    ref, ens_c = chaos_ensemble(b.take(sub), sub, iters, oracle)
    assert np.array_equal(ref["trace"], g["ref_trace"]) and np.array_equal(ref["hist"], g["ref_hist"], equal_nan=True)
    assert np.array_equal(ens_c["k_calm"], g["k_calm_c"]) and np.allclose(ens_c["spread"], g["spread_c"], rtol=1e-9, atol=0)
    ens = dict(k_calm=g["k_calm"], hist_spread=g["hist_spread"], spread=g["spread"])
    ok, rep = prefix_parity({k: got[k][sub] for k in ("X", "F", "hist", "trace")}, ref, ens)
    calm = ens["k_calm"] >= iters
    ratio = rep["err"][~calm] / np.maximum(ens["spread"][~calm], 1e-300)
    print("%s: %d sampled problems: %d calm max err %.2e | %d with a chaotic tail: calm prefixes %s ADMM iterations reproduced exactly; "
          "final distance / own ensemble spread median %.2f max %.2f"
          % (label, len(sub), calm.sum(), rep["err"][calm].max() if calm.any() else 0.0, (~calm).sum(), sorted(ens["k_calm"][~calm].tolist()),
             np.median(ratio) if ratio.size else 0.0, ratio.max() if ratio.size else 0.0))
    assert np.all(ok), {int(sub[i]): w for i, w in rep["why"].items()}
    assert np.all(rep["err"][calm] < calm_tol)
    assert np.array_equal(got["stats"][sub][calm], ref["stats"][calm])      # calm problems: the oracle's whole discrete path
    assert np.array_equal(got["stats"][sub][:, 5], ref["stats"][:, 5])
    if iters <= 10:
        assert np.array_equal(got["stats"][sub][:, 0], ref["stats"][:, 0])  # (at 100 iterations a chaotic problem may cross exit_tol an iteration apart)
    return ref, ens


def _go2_wb(B, first=0):
    go2 = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "go2.json")).read())
    wb = problems.make_wb_batch(go2, B, first=first, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    return go2, wb


@pytest.mark.parametrize("fused_direct", [16, 0], ids=["fused", "lockstep"])
def test_ddp_problems_that_hit_maxiter_follow_the_cpu_twin(fused_direct):
    """Problems 2, 13 and 14 of the bench's Go2 H = 60 batch never reach SolverDDP's stopping threshold (|Q_u|^2 < 1e-9) within
    its 100 iterations.  Not a regularisation limit cycle: the regularisation stays at its floor (1e-9) and the cost falls
    monotonically -- the Gauss-Newton DDP converges linearly with partial steps (alpha 1/16 ... 1/2) on these plans and simply
    runs out of iterations (EXPERIMENTS.md 9).  GPU and CPU twin must agree on every discrete decision along the way."""
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    from oracle import ik_oracle_c as ic
    from bunmpc_amd import _lib
    go2, wb = _go2_wb(16)
    kb = KinoDynDeviceBatch(wb, go2, num_iters=10)
    old = _lib.lib().bmpc_ik_set_fused_direct_max(fused_direct)     # 16: the whole batch inside the fused kernel; 0: the multi-kernel path
    try:
        kb.solve()
    finally:
        _lib.lib().bmpc_ik_set_fused_direct_max(old)
    g = kb.results()
    assert np.all(g["ik_fused_iters"] > 0) == (fused_direct > 0)
    r = ic.solve_wb_batch(ic.Model(go2), wb, g["X"], trace=True)
    long_runs = np.where(r["status"] == 1)[0]
    assert len(long_runs) >= 3 and {2, 13, 14} <= set(long_runs.tolist())
    assert np.array_equal(g["ik_iters"], r["iters"]) and np.array_equal(g["ik_status"], r["status"])
    for i in range(16):
        n = int(r["iters"][i])
        tg, tc = g["ik_trace"][i, :n], r["trace"][i, :n]
        assert np.array_equal(tg[:, 1], tc[:, 1]), i                     # regularisation after every iteration
        assert np.array_equal(tg[:, 2], tc[:, 2]), i                     # accepted step length of every iteration
        assert np.all(np.abs(tg[:, 0] - tc[:, 0]) <= 1e-6 * np.abs(tc[:, 0])), i     # cost after every iteration
        assert abs(g["ik_cost"][i] - r["cost"][i]) <= 1e-9 * abs(r["cost"][i])
        assert rel_l2(g["xs"][i].reshape(-1), r["xs"][i].reshape(-1)) < 1e-6
    for i in long_runs:
        tr = g["ik_trace"][i, :100]
        assert r["iters"][i] == 100 and g["ik_status"][i] == 1
        assert np.all(tr[:, 1] <= 1e-8)                                  # regularisation never leaves its floor: no limit cycle
        assert np.all(np.diff(tr[1:, 0]) <= 0) and np.all(tr[:, 2] > 0)  # every iteration accepts a step and lowers the cost
        assert tr[-1, 3] > 1e-9                                          # ... but the stopping criterion is not reached


def _centroidal_invariants(b, got, oracle, mu):
    B, H, E = b.B, b.H, b.E
    assert np.all(got["stats"][:, 5] == 0) and np.all(got["stats"][:, 0] == 10)
    F = got["F"].reshape(B, H, E, 3)
    assert np.all(F[b.cnt_plan[..., 0] == 0] == 0.0)                  # swing feet carry no force
    s = F[..., 0] ** 2 + F[..., 1] ** 2
    assert np.all(F[..., 2] >= 0) and np.all(s <= mu * F[..., 2] * (1 + 1e-6) + 1e-9)   # image of the reference's "SoC" map
    X = got["X"].reshape(B, H + 1, 9)
    assert np.all(np.isfinite(X)) and np.all(np.abs(X[:, 0, :3] - b.x_init[:, :3]) < 0.15)     # the CoM starts near the robot (a penalty, not a constraint)
    for i in np.arange(0, B, 512):
        A, bf = oracle.dense_A_f(b.cnt_plan[i], b.dt[i], b.m, got["F"][i], b.x_init[i])
        r = np.linalg.norm(A @ got["X"][i] - bf)
        assert abs(r - got["dyn_viol"][i]) <= 1e-4 * max(r, 1e-3)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_full_size_go2_bound_h40(oracle, precision):
    """BASELINE config 3 at its full size (B = 4096, H = 40) through the device-resident path: size-independent properties
    on every problem, oracle parity on a 64-problem sample.  (Synthetic Go2 runs with mu = 10, not the reference's mu = 1:
    the reference algorithm NaNs for a 15 kg robot at mu = 1, tests/test_oracle_cpu.py.)"""
    B = 4096
    b = problems.make_batch("go2_bound", B, H=40)
    dev = bb.DeviceBatch(b, num_iters=10, precision=precision, keep_hist=True)
    dev.solve()
    got = dev.results()
    _centroidal_invariants(b, got, oracle, b.mu)
    dev.solve()
    again = dev.results()
    for k in "XFP":
        assert np.array_equal(again[k], got[k])                       # deterministic reductions: bit-identical re-solve
    sub = np.arange(0, B, 64)
    if precision == "f64":
        _sampled_prefix_parity("go2_bound_h40", b, sub, 10, got, oracle, "go2_bound H=40 B=4096 f64")
    else:
        ref, spread = cpu_spread(b.take(sub), 10, oracle, with_numpy=False)
        err = np.maximum(rel_l2(got["X"][sub], ref["X"]), rel_l2(got["F"][sub], ref["F"]))
        assert np.array_equal(got["stats"][sub][:, [0, 5]], ref["stats"][:, [0, 5]])
        calm = spread <= 1e-9
        print("go2_bound H=40 B=4096 f32: sampled parity calm %d problems median %.2e max %.2e" % (calm.sum(), np.median(err[calm]), err[calm].max()))
        assert np.median(err[calm]) <= FP32_MEDIAN and np.all(err[calm] <= FP32_MAX)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_full_size_solo12_mixed(oracle, precision):
    """BASELINE config 4's per-GPU share (Solo12 trot / bound / pace mixed, H = 20, B = 4096 = 32768 / 8, per-problem weights):
    size-independent properties on every problem, a bit-identical second solve, oracle parity on a 64-problem sample inside the
    measured envelope (bound and pace sit in the reference algorithm's chaotic regime, tests/util.py)."""
    B = 4096
    b = problems.make_batch("solo12_mixed", B)
    assert b.W_X.shape[0] == B and len(set(b.gait_id.tolist())) == 3          # the per-problem-weights path, all three gaits
    dev = bb.DeviceBatch(b, num_iters=10, precision=precision, keep_hist=True)
    dev.solve()
    got = dev.results()
    assert bb._lib.lib().bmpc_biconvex_last_kernel_name() == (b"biconvex_admm_kernel" if precision == "f64" else b"biconvex_admm_kernel_f32")
    _centroidal_invariants(b, got, oracle, b.mu)
    dev.solve()
    again = dev.results()
    for k in "XFP":
        assert np.array_equal(again[k], got[k])
    sub = np.arange(0, B, 64)
    if precision == "f64":
        # problem 2304 is the one round 3's first run of this test went red on (1.2e-3 from the strict oracle where three CPU
        # restatements agreed to 8e-5): its ensemble leaves the common path in ADMM iteration 7 (the violation's range jumps from
        # 3e-15 to 1e-4 within that iteration's force FISTA, fista.cpp:38 / :52-70) and spreads to 1.3e-3 -- the GPU is one more
        # member.  Its first seven ADMM iterations are reproduced exactly.
        _sampled_prefix_parity("solo12_mixed", b, sub, 10, got, oracle, "solo12_mixed B=4096 f64")
    else:
        ref, spread = cpu_spread(b.take(sub), 10, oracle, with_numpy=False)
        err = np.maximum(rel_l2(got["X"][sub], ref["X"]), rel_l2(got["F"][sub], ref["F"]))
        assert np.array_equal(got["stats"][sub][:, [0, 5]], ref["stats"][:, [0, 5]])
        calm = spread <= 1e-9
        print("solo12_mixed B=4096 f32: sampled parity calm %d problems median %.2e max %.2e" % (calm.sum(), np.median(err[calm]), err[calm].max()))
        assert np.median(err[calm]) <= FP32_MEDIAN and np.all(err[calm] <= FP32_MAX)


def test_config2_batch_1024_through_the_default_dispatch(oracle):
    """BASELINE config 2 (Solo12 trot, H = 20, B = 1024 perturbed initial conditions, fp64) exactly as a caller gets it: no
    mapping override, so the dispatch takes the one-problem-per-wave kernel (asserted) -- invariants on all 1024, sampled oracle
    parity with the oracle's discrete path."""
    B = 1024
    b = problems.make_batch("solo12_trot", B)
    dev = bb.DeviceBatch(b, num_iters=10)
    dev.solve()
    assert bb._lib.lib().bmpc_biconvex_last_kernel_name() == b"biconvex_latency_kernel"
    got = dev.results()
    _centroidal_invariants(b, got, oracle, b.mu)
    sub = np.arange(0, B, 16)
    ref, spread = cpu_spread(b.take(sub), 10, oracle, with_numpy=False)
    err, bound = within_envelope({k: got[k][sub] for k in "XF"}, ref, spread)
    assert np.array_equal(got["stats"][sub], ref["stats"])
    print("solo12_trot B=1024 default dispatch: sampled parity median %.2e max %.2e" % (np.median(err), err.max()))
    assert np.all(err <= bound) and np.median(err) < 1e-12
    dev.solve()
    again = dev.results()
    for k in "XFP":
        assert np.array_equal(again[k], got[k])


def test_full_size_go2_h60_kinodyn(oracle):
    """BASELINE config 5's per-GPU share (synthetic Go2, trot, H = 60 / H_ik = 30, B = 1024): properties of every solution and
    twin parity on a sample."""
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    from oracle import ik_oracle_c as ic
    B = 1024
    go2, wb = _go2_wb(B)
    T, H = wb.ik_T, wb.dyn.H
    kb = KinoDynDeviceBatch(wb, go2, num_iters=10)
    kb.solve()
    g = kb.results()
    # centroidal part
    assert np.all(g["stats"][:, 5] == 0) and np.all(np.isfinite(g["X"]))
    F = g["F"].reshape(B, H, 4, 3)
    assert np.all(F[wb.dyn.cnt_plan[..., 0] == 0] == 0.0) and np.all(F[..., 2] >= 0)
    # whole-body part: statuses, stopping criterion, the trajectory is a rollout of its own controls from x0
    st = g["ik_status"]
    assert set(np.unique(st).tolist()) <= {0, 1} and np.all(g["ik_iters"][st == 1] == 100) and np.all(g["ik_stop"][st == 0] < 1e-9)
    assert 0.02 < (st == 1).mean() < 0.12                               # ~6 % run out of iterations (see the maxiter test above)
    xs, us, dt = g["xs"], g["us"], wb.dyn.dt[:, :T]
    assert np.array_equal(xs[:, 0], wb.x)
    v, a = xs[:, :-1, 19:], us
    assert np.abs(xs[:, 1:, 19:] - (v + a * dt[:, :, None])).max() < 1e-12                                  # v+ = v + a dt
    assert np.abs(xs[:, 1:, 7:19] - (xs[:, :-1, 7:19] + v[:, :, 6:] * dt[:, :, None] + a[:, :, 6:] * dt[:, :, None] ** 2)).max() < 1e-12
    assert np.abs(np.linalg.norm(xs[:, :, 3:7], axis=2) - 1.0).max() < 1e-12
    m = ic.Model(go2)
    for i in range(0, B, 128):          # base block of the Euler step through the CPU twin's integrate
        for t in range(0, T, 7):
            dx = np.concatenate([xs[i, t, 19:] * dt[i, t] + us[i, t] * dt[i, t] ** 2, us[i, t] * dt[i, t]])
            assert np.abs(ic.state_ops(m, xs[i, t], xs[i, t], dx)["xint"] - xs[i, t + 1]).max() < 1e-12
    # cost reported = cost of the returned trajectory, recomputed by the CPU twin's node model
    tasks = np.array(wb.ik_tasks)
    Xk = g["X"].reshape(B, H + 1, 9)[:, :T + 1]
    tasks[:, :, 21:24], tasks[:, :, 25:28], tasks[:, :, 28:31] = Xk[:, :, 0:3], wb.dyn.m * Xk[:, :, 3:6], Xk[:, :, 6:9]
    for i in range(0, B, 256):
        c = sum(ic.node(m, T, t, dt[i], tasks[i], wb.state_w[0], wb.x_reg[i], wb.ctrl_w[0], xs[i, t], us[i, t] if t < T else None)["cost"]
                for t in range(T + 1))
        assert abs(c - g["ik_cost"][i]) <= 1e-10 * abs(c)
    # a second solve is bit-identical
    kb.solve()
    g2 = kb.results()
    assert np.array_equal(g2["xs"], g["xs"]) and np.array_equal(g2["ik_iters"], g["ik_iters"]) and np.array_equal(g2["X"], g["X"])
    # sampled parity: the CPU twin on the GPU's own centroidal solution
    sub = np.arange(0, B, 64)
    r = ic.solve_wb_batch(m, wb.take(sub), g["X"][sub])
    assert np.array_equal(r["iters"], g["ik_iters"][sub]) and np.array_equal(r["status"], st[sub])
    assert np.all(np.abs(r["cost"] - g["ik_cost"][sub]) <= 1e-8 * np.abs(r["cost"]))
    e = rel_l2(g["xs"][sub].reshape(len(sub), -1), r["xs"].reshape(len(sub), -1))
    print("go2 H=60 B=1024: DDP iterations mean %.1f, not converged %d, sampled xs rel-L2 vs CPU twin max %.2e" % (g["ik_iters"].mean(), (st == 1).sum(), e.max()))
    assert np.all(e < 1e-6)
    # centroidal sample against the strict oracle, in the measured envelope
    ref, spread = cpu_spread(wb.dyn.take(sub[:8]), 10, oracle)
    err, bound = within_envelope({k: g[k][sub[:8]] for k in "XF"}, ref, spread)
    assert np.all(err <= bound), (err, bound)


def test_the_references_own_call_at_scale_100_admm_iterations(oracle):
    """kd.optimize(q, v, 100, 1) -- the call the reference's generator makes every 50 ms (abstract_cyclic_gen.py:663) -- as a
    batch of BASELINE's size: B = 4096 Solo12 trot, full KinoDynMP.optimize (centroidal ADMM at num_iters = 100 + whole-body
    IK-DDP).  At 100 iterations the ADMM's early exit (||A_f X - b_f|| < 1e-3, biconvex.cpp:111-114) makes the iteration count
    differ per problem, and a fifth of the problems reach the chaotic regime of the force FISTA (tests/util.py).  Invariants on
    every problem; on a 64-problem sample the prefix check against the strict oracle under the committed per-problem ensembles
    (the oracle's exact ADMM / FISTA counts on calm problems); the IK-DDP on the GPU's own centroidal solution against the
    compiled CPU twin."""
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    from oracle import ik_oracle_c as ic
    B, N = 4096, 100
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "solo12.json")).read())
    wb = problems.make_wb_batch(model, B)
    kb = KinoDynDeviceBatch(wb, model, num_iters=N, keep_hist=True)
    kb.solve()
    g = kb.results()
    H, T = wb.dyn.H, wb.ik_T
    n_admm = g["stats"][:, 0]
    assert np.all(g["stats"][:, 5] == 0) and np.all(np.isfinite(g["X"])) and np.all(np.isfinite(g["F"]))
    early = n_admm < N
    assert 0.5 < early.mean() and np.all(g["dyn_viol"][early] < 1e-3)     # the exit rule itself
    assert len(np.unique(n_admm)) > 10                                   # ... and it does make the counts differ per problem
    hist = g["hist"]
    for i in range(0, B, 97):                                            # the history: one row per iteration run, NaN after
        assert np.all(np.isfinite(hist[i, :n_admm[i]])) and np.all(np.isnan(hist[i, n_admm[i]:])) and hist[i, n_admm[i] - 1] == g["dyn_viol"][i]
        assert np.array_equal(g["trace"][i, n_admm[i] - 1], g["stats"][i, 1:5])
    F = g["F"].reshape(B, H, 4, 3)
    assert np.all(F[wb.dyn.cnt_plan[..., 0] == 0] == 0.0) and np.all(F[..., 2] >= 0)
    assert np.all(F[..., 0] ** 2 + F[..., 1] ** 2 <= wb.dyn.mu * F[..., 2] * (1 + 1e-6) + 1e-9)
    st = g["ik_status"]
    assert set(np.unique(st).tolist()) <= {0, 1} and np.all(g["ik_stop"][st == 0] < 1e-9) and (st == 1).mean() < 0.01
    assert np.array_equal(g["xs"][:, 0], wb.x)
    # sampled parity, centroidal part: x_init as the CPU twin computes it from (q, v) (kino_dyn.cpp:42)
    m = ic.Model(model)
    sub = np.arange(0, B, 64)
    wb.dyn.x_init[:] = ic.centroidal_state(m, wb.x)
    _sampled_prefix_parity("kinodyn_solo12_n100", wb.dyn, sub, N, g, oracle, "KinoDyn solo12 B=4096 num_iters=100", calm_tol=1e-10)
    # whole-body part: the CPU twin on the GPU's own centroidal solution
    r = ic.solve_wb_batch(m, wb.take(sub), g["X"][sub])
    assert np.array_equal(r["iters"], g["ik_iters"][sub]) and np.array_equal(r["status"], st[sub])
    assert np.all(np.abs(r["cost"] - g["ik_cost"][sub]) <= 1e-8 * np.abs(r["cost"]))
    e = rel_l2(g["xs"][sub].reshape(len(sub), -1), r["xs"].reshape(len(sub), -1))
    print("KinoDyn n100 B=4096: ADMM iterations min %d median %d max %d (%d%% exit early); DDP iterations mean %.1f; sampled xs rel-L2 vs CPU twin max %.2e"
          % (n_admm.min(), np.median(n_admm), n_admm.max(), 100 * early.mean(), g["ik_iters"].mean(), e.max()))
    assert np.all(e < 1e-6)
    kb.solve()
    g2 = kb.results()
    assert np.array_equal(g2["X"], g["X"]) and np.array_equal(g2["xs"], g["xs"]) and np.array_equal(g2["trace"], g["trace"])

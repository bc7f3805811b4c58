"""CPU tests of the oracle itself (no GPU): the C restatement against its independent
numpy/scipy twin, against hand-derived values, against the committed golden fixtures, and
the algorithm's own rounding sensitivity that bounds what "parity" can mean.

PARITY UNPINNED: the reference has no golden vectors for this path and cannot be built here."""
import glob
import os

import numpy as np
import pytest

from bunmpc_amd import problems
from oracle import oracle_np
from tests.util import rel_l2

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith(("ik_", "chaos_")))          # ik_*: whole-body fixtures (tests/test_ik_twin_cpu.py); chaos_*: ensembles (below)


def _np_solve(b, i, ref, iters):
    X0, F0, P0 = b.warm_start()
    return oracle_np.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], ref["Qx"][i], ref["qx"][i],
                                    ref["Qf"][i], ref["lbx"][i], ref["ubx"][i], X0[i], F0[i], P0[i],
                                    rho=b.rho, num_iters=iters, mu=b.mu)


def test_c_matches_numpy_trot(oracle):
    b = problems.make_batch("solo12_trot", 3)
    ref = oracle.solve_batch(b, num_iters=10)
    for i in range(b.B):
        rn = _np_solve(b, i, ref, 10)
        assert np.array_equal(rn["stats"], ref["stats"][i])
        for k in "XFP":
            assert rel_l2(ref[k][i], rn[k]) < 1e-12, k
        assert rn["L_x"] == ref["L_x"][i] and rn["L_f"] == ref["L_f"][i]


def test_c_matches_numpy_with_backtracking(oracle):
    """L0 far below the Lipschitz constants forces retries in both FISTA loops (fista.cpp:20-22)."""
    b = problems.make_batch("solo12_trot", 1)
    X0, F0, P0 = b.warm_start()
    pre = oracle.solve_batch(b, num_iters=0)
    kw = dict(L_x=1e4, L_f=10.0, rho=b.rho, num_iters=3)
    args = (b.cnt_plan[0], b.dt[0], b.m, b.x_init[0], pre["Qx"][0], pre["qx"][0], pre["Qf"][0],
            pre["lbx"][0], pre["ubx"][0], X0[0], F0[0], P0[0])
    rc = oracle.biconvex_solve(*args, **kw)
    rn = oracle_np.biconvex_solve(*args, **kw)
    assert rc["stats"][3] > 0 and rc["stats"][4] > 0
    assert np.array_equal(rc["stats"], rn["stats"])
    assert rc["L_x"] == rn["L_x"] and rc["L_f"] == rn["L_f"]
    for k in "XFP":
        assert rel_l2(rc[k], rn[k]) < 1e-11, k


def test_dense_matrices_match_numpy(oracle):
    rng = np.random.default_rng(7)
    b = problems.make_batch("solo12_trot", 1)
    H, E = b.H, b.E
    X = rng.standard_normal(9 * (H + 1))
    F = rng.standard_normal(3 * E * H)
    A, bx = oracle.dense_A_x(b.cnt_plan[0], b.dt[0], b.m, X)
    An, bn = oracle_np.build_A_x(X, b.cnt_plan[0], b.dt[0], b.m)
    assert np.allclose(A, An.toarray(), rtol=0, atol=1e-15) and np.allclose(bx, bn, rtol=0, atol=1e-15)
    A, bf = oracle.dense_A_f(b.cnt_plan[0], b.dt[0], b.m, F, b.x_init[0])
    An, bn = oracle_np.build_A_f(F, b.cnt_plan[0], b.dt[0], b.m, b.x_init[0])
    assert np.allclose(A, An.toarray(), rtol=0, atol=1e-14) and np.allclose(bf, bn, rtol=0, atol=1e-14)


def test_hand_derived_single_knot(oracle):
    """H = 1, one foot in contact: every entry of A_x, b_x, A_f, b_f from App. A.1/A.2 by hand."""
    m, dt = 2.0, 0.1
    cnt = np.zeros((1, 4, 4))
    cnt[0, 0] = [1, 0.2, -0.1, 0.0]       # foot 0 in contact at r
    cnt[0, 1] = [0, 9.0, 9.0, 9.0]        # swing foot: must contribute nothing
    X = np.arange(18, dtype=float) * 0.1  # com_0 = (0, .1, .2)
    A, b = oracle.dense_A_x(cnt, np.array([dt]), m, X)
    p = X[0:3] - cnt[0, 0, 1:4]
    assert np.allclose(A[3:6, 0:3], np.eye(3) * dt / m)
    assert np.allclose(A[6:9, 0:3], dt * np.array([[0, p[2], -p[1]], [-p[2], 0, p[0]], [p[1], -p[0], 0]]))
    assert np.all(A[:, 3:] == 0) and np.all(A[0:3] == 0) and np.all(A[9:] == 0)
    exp_b = np.zeros(18)
    exp_b[3:9] = X[12:18] - X[3:9]
    exp_b[5] += 9.81 * dt
    assert np.allclose(b, exp_b)
    F = np.zeros(12)
    F[0:3] = [1.0, 2.0, 3.0]
    F[3:6] = [7.0, 7.0, 7.0]              # swing foot force is masked by its flag
    x_init = np.arange(9, dtype=float)
    A, b = oracle.dense_A_f(cnt, np.array([dt]), m, F, x_init)
    assert np.allclose(A[0:9, 0:9] - np.eye(9), np.pad(dt * np.array([[0, -3, 2], [3, 0, -1], [-2, 1, 0.0]]), ((6, 0), (0, 6))))
    blk = -np.eye(9)
    blk[0:3, 3:6] = dt * np.eye(3)
    assert np.allclose(A[0:9, 9:18], blk)
    assert np.allclose(A[9:18, 0:9], np.eye(9)) and np.all(A[9:18, 9:18] == 0)
    r = cnt[0, 0, 1:4]
    exp = np.zeros(18)
    exp[3:6] = -F[0:3] * dt / m
    exp[5] += 9.81 * dt
    exp[6:9] = np.cross(F[0:3], r) * dt
    exp[9:18] = x_init
    assert np.allclose(b, exp)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    b = problems.make_batch(str(g["config"]), g["X"].shape[0])
    for k in ("cnt_plan", "dt", "x_init", "X_nom", "X_ter", "W_X", "W_X_ter", "W_F", "bounds"):
        assert np.array_equal(getattr(b, k), g[k]), "problem generator drifted: " + k
    r = oracle.solve_batch(b, num_iters=int(g["num_iters"]))
    assert np.array_equal(r["stats"], g["stats"])
    for k in "XFP":
        assert np.all(rel_l2(r[k], g[k]) < 1e-12), k


def test_golden_files_present():
    assert len(GOLDEN) >= 4


def test_invariants_of_a_trot_solve(oracle):
    b = problems.make_batch("solo12_trot", 4)
    r = oracle.solve_batch(b, num_iters=10)
    H, E = b.H, b.E
    F = r["F"].reshape(b.B, H, E, 3)
    flags = b.cnt_plan[..., 0]
    assert np.all(F[flags == 0] == 0.0)                       # swing feet never get a force
    s = F[..., 0] ** 2 + F[..., 1] ** 2
    assert np.all(F[..., 2] >= 0) and np.all(s <= b.mu * F[..., 2] * (1 + 1e-9) + 1e-12)  # fista.cpp:52-70 set
    X = r["X"].reshape(b.B, H + 1, 9)
    assert np.all(np.abs(X[:, 0] - b.x_init) < 2e-2)          # x_init rows are a penalty, not exact
    assert np.all(X[:, :H, 2] <= r["ubx"].reshape(b.B, H + 1, 9)[:, :H, 2] + 1e-12)
    assert np.all(r["stats"][:, 5] == 0)


def test_restatements_spread(oracle):
    """How far two faithful CPU restatements drift apart: nothing on trot (above), ~1e-4 on
    bound / pace once the force FISTA sits on the expansive branch of the "SoC" projection."""
    b = problems.make_batch("solo12_mixed", 3)          # gaits: pace, trot, bound
    ref = oracle.solve_batch(b, num_iters=10)
    errs = [rel_l2(ref["X"][i], _np_solve(b, i, ref, 10)["X"]) for i in range(3)]
    assert errs[1] < 1e-12                              # trot
    assert 1e-9 < errs[0] < 5e-3 and 1e-9 < errs[2] < 5e-3


def test_restatements_spread_at_100_iterations(oracle):
    b = problems.make_batch("solo12_trot", 5).slice(4, 5)
    ref = oracle.solve_batch(b, num_iters=100)
    rn = _np_solve(b, 0, ref, 100)
    assert abs(int(rn["stats"][0]) - int(ref["stats"][0, 0])) <= 1
    assert 1e-9 < rel_l2(ref["X"][0], rn["X"]) < 5e-3


def test_reference_projection_blows_up_for_go2_with_mu_1(oracle):
    """Why the synthetic Go2 config carries mu = 10: with the reference's fixed mu = 1 both
    restatements diverge to NaN in the first force step (status 2 = "solver diverged")."""
    b = problems.make_batch("go2_bound", 2)
    b.mu = 1.0
    r = oracle.solve_batch(b, num_iters=2)
    assert np.all(r["stats"][:, 5] == 2) and np.all(r["stats"][:, 0] == 1)
    assert np.isnan(r["F"]).any()
    b.mu = 10.0
    assert np.all(oracle.solve_batch(b, num_iters=2)["stats"][:, 5] == 0)


def test_gait_phase_functions(oracle):
    lib = oracle.lib()
    rng = np.random.default_rng(3)
    for g in (problems.TROT, problems.BOUND, problems.JUMP, problems.PACE):
        for t in np.round(rng.uniform(0, 3, 40), 3):
            for j in range(4):
                sp, off = g.stance_percent[j], g.phase_offset[j]
                assert lib.orc_gait_phase(t, g.gait_period, sp, off) == int(problems.gait_phase(t, g.gait_period, sp, off))
                assert lib.orc_gait_percent_in_phase(t, g.gait_period, sp, off) == pytest.approx(
                    float(problems.gait_percent_in_phase(t, g.gait_period, sp, off)), abs=1e-15)
    # trot at t = 0: FL/HR (offset 0) in stance, FR/HL (offset .5, phi = .25 <= .3) in stance too
    assert [lib.orc_gait_phase(0.0, 0.5, 0.6, o) for o in (0.0, 0.5, 0.5, 0.0)] == [1, 1, 1, 1]
    # phi == stance_time within 1e-4 counts as stance (gait_planner.cpp:49)
    assert lib.orc_gait_phase(0.30005, 0.5, 0.6, 0.0) == 1 and lib.orc_gait_phase(0.3002, 0.5, 0.6, 0.0) == 0


def test_matrix_free_cpu_variant_agrees_with_the_strict_restatement(oracle):
    """oracle/biconvex_fast.c (bench.py's second CPU baseline): same iterates and the same discrete path as the
    explicit-Hessian restatement wherever the iteration is not chaotic (trot), the usual envelope elsewhere"""
    b = problems.make_batch("solo12_trot", 24)
    a = oracle.solve_batch(b, num_iters=10)
    f = oracle.solve_batch(b, num_iters=10, fast=True)
    assert np.array_equal(a["stats"], f["stats"])
    for k in "XFP":
        assert np.all(rel_l2(f[k], a[k]) < 1e-12), k
    b = problems.make_batch("solo12_mixed", 12)
    a = oracle.solve_batch(b, num_iters=10)
    f = oracle.solve_batch(b, num_iters=10, fast=True)
    assert np.array_equal(a["stats"][:, [0, 5]], f["stats"][:, [0, 5]])
    assert np.all(rel_l2(f["X"], a["X"]) < 5e-3)


def test_per_iteration_history_agrees_between_the_restatements(oracle):
    """hist / trace (dynamics violation and running FISTA iteration / retry counts after every ADMM iteration): the strict C
    oracle, the matrix-free C variant and the numpy twin leave the same discrete path and the same violations on calm problems;
    the totals of the last row are the stats."""
    b = problems.make_batch("solo12_trot", 4)
    ref = oracle.solve_batch(b, num_iters=10, trace=True)
    fast = oracle.solve_batch(b, num_iters=10, trace=True, fast=True)
    assert np.array_equal(ref["trace"], fast["trace"])
    assert np.allclose(ref["hist"], fast["hist"], rtol=1e-10, atol=0)
    for i in range(b.B):
        n = int(ref["stats"][i, 0])
        assert np.array_equal(ref["trace"][i, n - 1], ref["stats"][i, [1, 2, 3, 4]]) and np.all(ref["trace"][i, n:] == -1)
        assert np.all(np.diff(ref["trace"][i, :n], axis=0) >= 0)
        rn = _np_solve(b, i, ref, 10)
        assert np.array_equal(rn["trace"], ref["trace"][i, :n]) and np.allclose(rn["hist"], ref["hist"][i, :n], rtol=1e-10, atol=0)


@pytest.mark.parametrize("name,config,H,iters", [("solo12_mixed", "solo12_mixed", None, 10), ("go2_bound_h40", "go2_bound", 40, 10)])
def test_committed_chaos_ensembles_reproduce(oracle, name, config, H, iters):
    """tests/golden/chaos_<name>.npz (tools/chaos_ensemble.py): the C members of the per-problem ensembles, re-run here on every
    fourth sampled problem, give the committed k_calm / spreads / strict history exactly -- the perturbations are a function of
    (seed, problem index, member) alone.  And the file tells what the ensembles are for: problem 2304 of solo12_mixed (the one
    round 3's full-size test went red on) leaves the common path in ADMM iteration 7 and spreads to ~1e-3 on the CPU alone."""
    from tests.util import chaos_ensemble
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "chaos_%s.npz" % name))
    pick = np.arange(0, len(g["sub"]), 4)
    if name == "solo12_mixed":
        i2304 = int(np.where(g["sub"] == 2304)[0][0])
        assert g["k_calm"][i2304] == 7 and 5e-4 < g["spread"][i2304] < 5e-3
        assert np.all(g["hist_spread"][i2304, :7] < 1e-13) and g["hist_spread"][i2304, 7] > 1e-5
        pick = np.union1d(pick, [i2304])
    sub = g["sub"][pick]
    b = problems.make_batch(config, 4096, H=H) if H else problems.make_batch(config, 4096)
    ref, ens = chaos_ensemble(b.take(sub), sub, iters, oracle)
    assert np.array_equal(ref["trace"], g["ref_trace"][pick]) and np.array_equal(ref["hist"], g["ref_hist"][pick], equal_nan=True)
    assert np.array_equal(ens["k_calm"], g["k_calm_c"][pick])
    assert np.allclose(ens["spread"], g["spread_c"][pick], rtol=1e-9, atol=0)
    assert np.allclose(ens["hist_spread"], g["hist_spread_c"][pick], rtol=1e-9, atol=0)
    # the numpy twin can only widen an ensemble
    assert np.all(g["k_calm"] <= g["k_calm_c"]) and np.all(g["spread"] >= g["spread_c"])

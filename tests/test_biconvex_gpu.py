"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the
C-ABI, against the CPU oracle (oracle/, test infrastructure) on identical seeded inputs.

Tolerance: north_star asks for trajectories within 1e-5 relative L2 of the CPU reference in
fp64; the kernel is matrix-free with FMA contraction while the oracle forms the explicit
Hessian without FMA, so agreement is at rounding level, not bitwise.  PARITY UNPINNED: the
oracle itself is pinned only by its numpy twin (no reference golden vectors exist)."""
import numpy as np
import pytest

from bunmpc_amd import batch as bb
from bunmpc_amd import problems
from tests.util import cpu_spread, rel_l2, within_envelope

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_lane_exchange_selftest(hiplib):
    from bunmpc_amd import _lib
    _lib.check(hiplib.bmpc_selftest_lanes())


@pytest.mark.parametrize("config,B,iters", [("solo12_trot_nominal", 1, 10), ("solo12_trot", 16, 10),
                                            ("solo12_mixed", 12, 1), ("go2_bound", 6, 2)])
def test_batch_matches_oracle(oracle, mapping, config, B, iters):
    """Runs that stay out of the chaotic regime described in test_chaotic_envelope (trot at the
    benchmark's 10 ADMM iterations; bound / pace / Go2 over their first 1-2): GPU within 1e-5
    rel-L2 of the strict CPU restatement (measured ~1e-15) and on the identical discrete path
    (iteration and retry counts)."""
    mapping("batch")
    b = problems.make_batch(config, B)
    ref = oracle.solve_batch(b, num_iters=iters)
    got = bb.solve_host(b, num_iters=iters)
    for k in ("X", "F", "P"):
        err = rel_l2(got[k], ref[k])
        assert np.all(err < TOL), (k, err)
    assert np.array_equal(got["stats"], ref["stats"])
    print(config, "max rel err", {k: float(rel_l2(got[k], ref[k]).max()) for k in "XFP"})


@pytest.mark.parametrize("config,B", [("solo12_mixed", 12), ("go2_bound", 6)])
def test_chaotic_envelope(oracle, config, B):
    """The reference's "SoC" projection (fista.cpp:52-70) uses the SQUARED tangential norm; on
    its cone branch (s > mu z) it is expansive once |f_xy| > ~0.5 N, the force FISTA then never
    converges (G stays ~1e-2 for all 150 iterations) and rounding differences grow ~x1.12 per
    iteration.  Bound / pace problems enter that regime after a few ADMM iterations: the CPU
    restatements (same formulas, different summation order) then differ by 1e-4 ... 4e-3 rel-L2.
    The GPU is held, problem by problem, to 10 x the spread the three CPU restatements show on that
    very problem (tests/util.py: measured ratio <= 2.7), to 1e-5 where they agree, to the same ADMM
    count / status, and to the solution invariants (test_full_size_*)."""
    b = problems.make_batch(config, B)
    ref, spread = cpu_spread(b, 10, oracle)
    got = bb.solve_host(b, num_iters=10)
    assert np.array_equal(got["stats"][:, [0, 5]], ref["stats"][:, [0, 5]])
    err, bound = within_envelope(got, ref, spread)
    assert np.all(err <= bound), (err, bound)
    print(config, "GPU-C rel err", err, "CPU spread", spread)


def test_hundred_admm_iterations(oracle):
    """The reference's own call is kd.optimize(q, v, 100, 1) (abstract_cyclic_gen.py:663).  Over
    ~60-100 ADMM iterations the algorithm amplifies rounding-order differences: the CPU
    restatements differ by up to ~5e-4 rel-L2 here and can exit one ADMM iteration apart when
    ||dyn|| crosses exit_tol = 1e-3 within rounding (tests/test_oracle_cpu.py::
    test_restatements_spread_at_100_iterations).  The GPU is held to 10 x the CPU spread measured on
    each problem (1e-5 where the CPU restatements agree), plus the exit condition itself."""
    b = problems.make_batch("solo12_trot", 5)
    ref, spread = cpu_spread(b, 100, oracle)
    got = bb.solve_host(b, num_iters=100)
    assert np.all(np.abs(got["stats"][:, 0] - ref["stats"][:, 0]) <= 1)
    err, bound = within_envelope(got, ref, spread)
    assert np.all(err <= bound), (err, bound)
    done = got["stats"][:, 0] < 100
    assert np.all(got["dyn_viol"][done] < 1e-3)
    same = np.all(got["stats"] == ref["stats"], axis=1)
    assert np.all(same[spread < 1e-9])             # where the CPU restatements agree the discrete path is identical
    print("100 iters: GPU-C rel err", err, "CPU spread", spread, "same discrete path:", same)


GOLDEN = sorted(p for p in __import__("glob").glob(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "*.npz"))
                if not __import__("os").path.basename(p).startswith(("ik_", "chaos_")))   # ik_*: whole-body fixtures (tests/test_ik_gpu.py); chaos_*: CPU ensembles


@pytest.mark.parametrize("path", GOLDEN, ids=[__import__("os").path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(path):
    """Committed inputs/outputs (tests/golden/make_golden.py); no oracle build needed."""
    g = np.load(path)
    b = problems.make_batch(str(g["config"]), g["X"].shape[0])
    got = bb.solve_host(b, num_iters=int(g["num_iters"]))
    assert np.array_equal(got["stats"], g["stats"])
    for k in ("X", "F", "P"):
        assert np.all(rel_l2(got[k], g[k]) < TOL), k
    assert np.array_equal(got["L_x"], g["L_x"]) and np.array_equal(got["L_f"], g["L_f"])


def _drive_handle(mp, b, i, iters, warm=True):
    for t in range(b.H):
        mp.set_contact_plan(b.cnt_plan[i, t], b.dt[i, t])
    mp.create_bound_constraints(b.bounds[0], 15.0, 15.0, 15.0)
    mp.create_cost_X(b.W_X[0], b.W_X_ter[0], b.X_ter[i], b.X_nom[i])
    mp.create_cost_F(b.W_F[0])
    if warm:
        X0, F0, P0 = b.warm_start()
        mp.set_warm_start_vars(X0[i], F0[i], P0[i])
    mp.optimize(b.x_init[i], iters)


def test_dropin_handle_matches_oracle(oracle):
    """biconvex_mpc_cpp.BiconvexMP driven exactly as abstract_cyclic_gen.py:391,611-614,663 does."""
    from bunmpc_amd.biconvex_mpc_cpp import BiconvexMP
    b = problems.make_batch("solo12_trot", 3)
    ref = oracle.solve_batch(b, num_iters=10)
    for i in range(b.B):
        mp = BiconvexMP(b.m, b.H, b.E)
        mp.set_rho(b.rho)
        mp.collect_statistics()
        _drive_handle(mp, b, i, 10)
        assert rel_l2(mp.return_opt_x(), ref["X"][i]) < TOL
        assert rel_l2(mp.return_opt_f(), ref["F"][i]) < TOL
        assert rel_l2(mp.return_opt_p(), ref["P"][i]) < TOL
        X = ref["X"][i].reshape(-1, 9)
        assert np.allclose(mp.return_opt_com(), X[:, 0:3], rtol=1e-9, atol=1e-12)
        assert np.allclose(mp.return_opt_mom(), np.hstack([b.m * X[:, 3:6], X[:, 6:9]]), rtol=1e-9, atol=1e-12)
        assert np.array_equal(mp.last_stats(), ref["stats"][i])
        assert len(mp.return_dyn_viol_hist()) == 10


def test_handle_state_persists_between_solves(oracle):
    """fista L_ is never reset and X/F/P survive (App. A.7): two optimize calls on one handle
    equal the oracle run twice with carried state."""
    from bunmpc_amd.biconvex_mpc_cpp import BiconvexMP
    b = problems.make_batch("solo12_trot", 1)
    mp = BiconvexMP(b.m, b.H, b.E)
    mp.set_rho(b.rho)
    mp.set_step_constants(2e5, 40.0)          # low enough to force retries that must persist
    _drive_handle(mp, b, 0, 3)
    L1 = mp.step_constants()
    _drive_handle(mp, b, 0, 3, warm=False)     # continues from the previous X/F/P
    pre = oracle.solve_batch(b, num_iters=0)
    X0, F0, P0 = b.warm_start()
    args = (b.cnt_plan[0], b.dt[0], b.m, b.x_init[0], pre["Qx"][0], pre["qx"][0], pre["Qf"][0],
            pre["lbx"][0], pre["ubx"][0])
    r1 = oracle.biconvex_solve(*args, X0[0], F0[0], P0[0], L_x=2e5, L_f=40.0, rho=b.rho, num_iters=3)
    assert (r1["L_x"], r1["L_f"]) == L1 and r1["stats"][3] > 0 and r1["stats"][4] > 0
    r2 = oracle.biconvex_solve(*args, r1["X"], r1["F"], r1["P"], L_x=r1["L_x"], L_f=r1["L_f"],
                               rho=b.rho, num_iters=3)
    assert rel_l2(mp.return_opt_x(), r2["X"]) < TOL and rel_l2(mp.return_opt_f(), r2["F"]) < TOL
    assert mp.step_constants() == (r2["L_x"], r2["L_f"])


@pytest.mark.parametrize("which", ["batch", "wave"])
def test_raw_form_with_backtracking(oracle, mapping, which):
    """raw cost/bound arrays + warm start + per-problem L0 (forces retries in both FISTA loops)."""
    mapping(which)
    b = problems.make_batch("solo12_trot", 6)
    pre = oracle.solve_batch(b, num_iters=0)
    raw = {k: pre[k] for k in ("Qx", "qx", "lbx", "ubx", "Qf")}
    Lx = np.array([2.25e6, 1e4, 1e5, 3e5, 2.25e6, 5e4])
    Lf = np.array([506.25, 10.0, 50.0, 506.25, 20.0, 100.0])
    X0, F0, P0 = b.warm_start()
    got = bb.solve_host(b, num_iters=3, raw=raw, warm=(X0, F0, P0), L_x=Lx, L_f=Lf)
    for i in range(b.B):
        r = oracle.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], pre["Qx"][i], pre["qx"][i],
                                  pre["Qf"][i], pre["lbx"][i], pre["ubx"][i], X0[i], F0[i], P0[i],
                                  L_x=Lx[i], L_f=Lf[i], rho=b.rho, num_iters=3)
        assert np.array_equal(got["stats"][i], r["stats"]), i
        assert got["L_x"][i] == r["L_x"] and got["L_f"][i] == r["L_f"]
        for k in "XFP":
            assert rel_l2(got[k][i], r[k]) < TOL, (i, k)
    assert got["stats"][:, 3:5].sum() > 0


@pytest.mark.parametrize("H", [3, 15, 16, 31, 32, 63])
def test_horizons_and_ragged_batches(oracle, mapping, H):
    """All three lane layouts (16/32/64 lanes per problem) of the one-knot-per-lane kernel, the H+1 == lanes edge, and batch
    sizes that do not fill the last wave."""
    mapping("batch")
    B = 7
    b = problems.make_batch("solo12_trot", B, H=H)
    iters = 2 if H < 40 else 1       # long horizons enter the chaotic regime (test_chaotic_envelope) sooner
    ref, spread = cpu_spread(b, iters, oracle, with_numpy=H >= 40)
    got = bb.solve_host(b, num_iters=iters)
    assert np.array_equal(got["stats"], ref["stats"])
    err, bound = within_envelope(got, ref, spread)
    if H < 40:
        assert np.all(err < TOL) and np.all(rel_l2(got["P"], ref["P"]) < TOL), err
    else:   # 3 s horizons amplify rounding already inside the first ADMM iteration: 10 x the CPU spread of each problem
        assert np.median(err) < 1e-8 and np.all(err <= bound), (err, bound)


@pytest.mark.parametrize("H,B", [(64, 5), (100, 3), (127, 4), (128, 3), (200, 3), (255, 2)])
def test_long_horizons_one_problem_per_workgroup(oracle, hiplib, H, B):
    """Horizons of 64 .. 255 knots (the reference's own sweep of solve times goes to 10 s: examples/analysis/solve_times_test.py) run one
    problem per workgroup of two / four waves, the neighbour exchanges and the segment sums crossing the wave boundaries through
    LDS.  Against the CPU oracle: the ADMM and FISTA counts and the iterates inside each problem's measured CPU spread (horizons
    this long are in the chaotic regime from the first ADMM iteration on: tests/util.py), the harness form and the raw form."""
    b = problems.make_batch("solo12_trot", B, H=H)
    ref, spread = cpu_spread(b, 1, oracle, with_numpy=False)
    got = bb.solve_host(b, num_iters=1, keep_hist=True)
    assert hiplib.bmpc_biconvex_last_kernel_name().decode() == "biconvex_admm_wg_kernel" and hiplib.bmpc_biconvex_last_lanes_per_problem() == (128 if H + 1 <= 128 else 192 if H + 1 <= 192 else 256)
    err, bound = within_envelope(got, ref, spread)
    print("H=%d: err %s bound %s spread %s" % (H, err, bound, spread))
    assert np.all(err <= bound), (err, bound)
    calm = spread <= 1e-9
    assert np.array_equal(got["stats"][calm], ref["stats"][calm])
    assert np.all(got["stats"][:, 0] == 1) and np.all(np.isfinite(got["X"]))
    # the force step alone (one ADMM iteration ends with the motion step: compare the FISTA counts of both)
    assert np.all(np.abs(got["stats"][:, 1:3] - ref["stats"][:, 1:3]) <= np.where(calm, 0, 150)[:, None])


@pytest.mark.parametrize("H,B,raw", [(64, 3, False), (65, 2, False), (127, 3, False), (128, 2, True), (129, 2, False), (191, 2, False), (192, 2, True), (255, 2, False)])
def test_long_horizons_before_the_chaos_sets_in(oracle, hiplib, H, B, raw):
    """The same kernels with FISTA capped at ten iterations per step and two ADMM iterations: too few for the expansive projection to
    amplify rounding, so EVERY problem must agree with the strict CPU oracle to rounding -- iterates, multipliers, step constants,
    every count, the violation history -- whichever wave boundary a knot sits at (knots 63 | 64, 127 | 128, 191 | 192) and with
    partly filled last waves.  Harness form and raw form."""
    b = problems.make_batch("solo12_trot", B, H=H)
    kw = dict(num_iters=2, maxit=10)
    if not raw:
        ref = oracle.solve_batch(b, **kw)
        got = bb.solve_host(b, keep_hist=True, **kw)
    else:
        pre = oracle.solve_batch(b, num_iters=0)
        rng = np.random.default_rng(H)
        rawd = {k: pre[k] for k in ("Qx", "qx", "lbx", "ubx", "Qf")}
        rawd["qf"] = rng.normal(0.0, 1e-3, pre["Qf"].shape)
        X0, F0, P0 = b.warm_start()
        got = bb.solve_host(b, keep_hist=True, raw=rawd, warm=(X0, F0, P0), **kw)
        rs = [oracle.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], rawd["Qx"][i], rawd["qx"][i], rawd["Qf"][i], rawd["lbx"][i], rawd["ubx"][i],
                                    X0[i], F0[i], P0[i], rho=b.rho, qf=rawd["qf"][i], **kw) for i in range(B)]
        ref = {k: np.stack([np.asarray(r[k]) for r in rs]) for k in ("X", "F", "P", "stats", "L_x", "L_f")}
    assert hiplib.bmpc_biconvex_last_kernel_name().decode() == "biconvex_admm_wg_kernel"
    assert np.array_equal(got["stats"], ref["stats"])
    for k in ("X", "F", "P"):
        assert np.all(rel_l2(got[k], ref[k]) < 1e-10), (k, rel_l2(got[k], ref[k]))
    assert np.allclose(got["L_x"], ref["L_x"], rtol=1e-14) and np.allclose(got["L_f"], ref["L_f"], rtol=1e-14)


def test_unsupported_shapes_are_refused():
    from bunmpc_amd import _lib
    b = problems.make_batch("solo12_trot", 1, H=256)
    with pytest.raises(_lib.BmpcError) as e:
        bb.solve_host(b, num_iters=1)
    assert e.value.code == _lib.BAD_ARG
    assert bb.solve_host(problems.make_batch("solo12_trot", 1).slice(0, 0), num_iters=1)["X"].shape[0] == 0


def test_fp32_kernels_use_no_scratch_memory(hiplib):
    """Two waves per SIMD cap the fp32 kernels at 256 registers; built with the SLP vectoriser they spilled 40-60 values per lane
    (345 MB of HBM traffic per launch for 75 MB of data).  The loaded code object must report no private segment."""
    assert hiplib.bmpc_biconvex_fp32_scratch_bytes() == 0


def test_fp32_variant_on_a_ragged_batch():
    """The fp32 kernels are a translation unit of their own (biconvex_admm_f32.hip): the padding problems of a last, partly
    filled wave (B = 7 at four problems per wave) read the wave's first problem and write nothing."""
    b = problems.make_batch("solo12_trot", 7, H=8)
    d64 = bb.solve_host(b, num_iters=3)
    d32 = bb.solve_host(b, num_iters=3, precision="f32")
    assert np.array_equal(d32["stats"][:, [0, 5]], d64["stats"][:, [0, 5]])
    for k in ("X", "F"):
        assert np.median(rel_l2(d32[k], d64[k])) < 1e-4 and rel_l2(d32[k], d64[k]).max() < 5e-3


@pytest.mark.parametrize("which", ["batch", "batch3", "wave"])
def test_diverging_problem_does_not_poison_neighbours(oracle, hiplib, mapping, which):
    """NaN handling (biconvex.cpp:106-109): a problem that blows up reports status 2 and NaNs;
    the problems sharing its wave (one with 32-lane segments, two with 21-lane segments: "batch3") are bit-for-bit what they are
    when solved without that neighbour."""
    mapping("wave" if which == "wave" else "batch")
    old3 = hiplib.bmpc_set_three_per_wave(1 if which == "batch3" else 0)
    try:
        _diverging_problem_body(oracle, hiplib, which)
    finally:
        hiplib.bmpc_set_three_per_wave(old3)


def _diverging_problem_body(oracle, hiplib, which):
    b = problems.make_batch("solo12_trot", 4)
    bad = problems.make_batch("solo12_trot", 4)
    bad.x_init[1, 2] = 1e200                      # overflow -> inf/NaN in the first gradient
    bad.X_nom[1] = 1e200
    got = bb.solve_host(bad, num_iters=4)
    assert hiplib.bmpc_biconvex_last_lanes_per_problem() == {"batch": 32, "batch3": 21, "wave": 0}[which]
    clean = bb.solve_host(b, num_iters=4)
    assert got["stats"][1, 5] == 2 and got["stats"][1, 0] == 1
    assert not np.isfinite(got["X"][1]).all()
    for i in (0, 2, 3):
        assert got["stats"][i, 5] == 0
        for k in "XFP":
            assert np.array_equal(got[k][i], clean[k][i]), (i, k)
    ref = oracle.solve_batch(bad.slice(1, 2), num_iters=4)
    assert ref["stats"][0, 5] == 2 and ref["stats"][0, 0] == 1


@pytest.mark.parametrize("which,H,precision", [("batch", 40, "f64"), ("batch", 40, "f32"), ("batch", 20, "f64"), ("batch", 20, "f32"),
                                               ("wave", 20, "f64"), ("batch", 15, "f64")])
def test_go2_at_the_references_mu_1_diverges_as_the_oracle(oracle, mapping, which, H, precision):
    """The reference's friction coefficient is fixed at mu = 1 (fista.hpp:60; the setter is not bound,
    srcpy/motion_planner/biconvex.cpp:19-44).  For the 15 kg synthetic Go2 its squared-norm "SoC" projection (fista.cpp:52-70) is
    expansive enough that the force FISTA overflows to NaN within the first or second ADMM iteration: BiConvexMP::optimize prints
    "solver diverged" and returns with the NaNs in place (biconvex.cpp:106-109).  The NATURAL divergence, through both centroidal
    kernels (one knot per lane with 1 / 2 / 4 problems per wave, one problem per wave) and both precisions: status 2, the
    oracle's ADMM count (1 or 2), the oracle's iteration counters and its NaN pattern in F -- and contact-free neighbours in the
    same batch (every fifth problem: no force, nothing for the projection to act on) untouched: bit for bit what they are in a
    batch at mu = 10."""
    mapping(which)
    B = 40
    b = problems.make_batch("go2_bound", B, H=H)
    b.cnt_plan[::5, :, :, 0] = 0.0
    free = np.zeros(B, bool)
    free[::5] = True
    b.mu = 1.0
    ref = oracle.solve_batch(b, num_iters=10)
    assert np.all(ref["stats"][~free, 5] == 2) and set(ref["stats"][~free, 0].tolist()) <= {1, 2, 3} and np.all(ref["stats"][free, 5] == 0)
    got = bb.solve_host(b, num_iters=10, precision=precision)
    name = bb._lib.lib().bmpc_biconvex_last_kernel_name()
    assert name == {"wave": b"biconvex_latency_kernel", "batch": b"biconvex_admm_kernel" if precision == "f64" else b"biconvex_admm_kernel_f32"}[which]
    assert np.array_equal(got["stats"][:, 5], ref["stats"][:, 5])                  # status 2 exactly where the oracle diverges
    if precision == "f64":
        assert np.array_equal(got["stats"], ref["stats"])                          # ... after the same ADMM / FISTA iterations
        assert np.array_equal(np.isnan(got["F"]), np.isnan(ref["F"]))              # ... leaving the NaNs where the oracle leaves them
    else:
        assert np.all(got["stats"][~free, 0] <= 3)
    assert np.all(np.isnan(got["F"][~free]).any(axis=1)) and np.all(np.isnan(got["dyn_viol"][~free]))
    b.mu = 10.0
    calm = bb.solve_host(b, num_iters=10, precision=precision)
    assert np.all(calm["stats"][:, 5] == 0)
    for k in "XFP":
        assert np.array_equal(got[k][free], calm[k][free]), k
    assert np.array_equal(got["stats"][free], calm["stats"][free]) and np.all(got["F"][free] == 0.0)


@pytest.mark.parametrize("config,B,H,iters", [("solo12_trot", 100, None, 10), ("solo12_mixed", 31, None, 10), ("solo12_trot", 7, 16, 3),
                                              ("solo12_trot", 64, 18, 4), ("go2_bound", 5, 20, 2)])
def test_three_problems_per_wave_equal_two_per_wave(oracle, hiplib, mapping, config, B, H, iters):
    """Horizons of 17..21 knots run three problems per wave in 21-lane segments (bmpc_set_three_per_wave, the default) instead of
    two in 32-lane segments.  The iterates of a problem do not depend on the mapping: only the segment sums behind the step
    decisions are added in another order, and wherever no decision sits within rounding of its threshold the two mappings leave
    the SAME BITS -- all of X, F, P, the step constants, the history, every count.  Batch sizes that are no multiple of three,
    partly filled last waves, per-problem weights (solo12_mixed), horizons below 21 knots (idle lanes inside the segments)."""
    mapping("batch")
    b = problems.make_batch(config, B, H=H) if H else problems.make_batch(config, B)
    assert 17 <= b.H + 1 <= 21
    out = {}
    for on in (1, 0):
        old = hiplib.bmpc_set_three_per_wave(on)
        try:
            out[on] = bb.solve_host(b, num_iters=iters, keep_hist=True)
            assert hiplib.bmpc_biconvex_last_lanes_per_problem() == (21 if on else 32)
        finally:
            hiplib.bmpc_set_three_per_wave(old)
    same = np.all(out[1]["trace"] == out[0]["trace"], axis=(1, 2))
    print("%s B=%d H=%d: %d of %d problems on the same discrete path in both mappings" % (config, B, b.H, same.sum(), B))
    assert same.mean() >= (0.6 if config == "solo12_mixed" else 0.9)      # (bound / pace: the chaotic regime, where any rounding difference flips counts)
    for k in ("X", "F", "P", "L_x", "L_f"):
        assert np.array_equal(out[1][k][same], out[0][k][same]), k
    for k in ("hist", "dyn_viol"):      # (segment sums themselves: the same terms added in another order)
        assert np.allclose(out[1][k][same], out[0][k][same], rtol=1e-12, atol=0, equal_nan=True), k
    assert np.array_equal(out[1]["stats"][same], out[0]["stats"][same])
    ref, spread = cpu_spread(b, iters, oracle, with_numpy=False)      # (chaotic problems: the measured envelope, tests/util.py)
    err, bound = within_envelope(out[1], ref, spread)
    assert np.all(err <= bound), (err, bound)
    calm = spread <= 1e-9
    assert np.array_equal(out[1]["stats"][calm], ref["stats"][calm])


@pytest.mark.parametrize("config,B,H,iters,warm", [("solo12_trot", 100, None, 10, False), ("solo12_mixed", 31, None, 10, False), ("solo12_trot", 9, 12, 4, False),
                                                   ("solo12_trot", 5, 28, 3, False), ("go2_bound", 6, 40, 3, False), ("solo12_trot", 33, None, 3, True),
                                                   ("solo12_trot", 3, 100, 2, False), ("solo12_trot", 2, 200, 1, True)])
def test_two_waves_per_simd_build_is_bit_identical(hiplib, mapping, config, B, H, iters, warm):
    """The fp64 batch kernel has a second build for two waves per SIMD (256 registers: x_k of the FISTA loops and its affine image
    rest in LDS between the iterations, bmpc_set_two_waves_per_simd).  Same operations in the same order: EVERY output bit for bit
    as from the one-wave build -- iterates, step constants, violation history, every per-iteration count -- with 16 / 32 / 64
    lanes per problem, two / four waves per problem (horizons of 100 and 200 knots), partly filled waves, the chaotic mixed-gait batch, warm starts with carried step constants."""
    mapping("batch")
    b = problems.make_batch(config, B, H=H) if H else problems.make_batch(config, B)
    old3 = hiplib.bmpc_set_three_per_wave(0)
    out = {}
    try:
        for mode in (1, 0):
            old = hiplib.bmpc_set_two_waves_per_simd(mode)
            try:
                r = bb.solve_host(b, num_iters=iters, keep_hist=True)
                assert hiplib.bmpc_biconvex_last_waves_per_simd() == (2 if mode else 1)
                if warm:      # a second call of the same solver objects: iterates and step constants carried over
                    r = bb.solve_host(b, num_iters=iters, keep_hist=True, warm=(r["X"], r["F"], r["P"]), L_x=r["L_x"], L_f=r["L_f"])
                out[mode] = r
            finally:
                hiplib.bmpc_set_two_waves_per_simd(old)
    finally:
        hiplib.bmpc_set_three_per_wave(old3)
    for k in ("X", "F", "P", "L_x", "L_f", "stats", "trace"):
        assert np.array_equal(out[1][k], out[0][k]), k
    for k in ("hist", "dyn_viol"):
        assert np.array_equal(out[1][k], out[0][k], equal_nan=True), k


def test_three_per_wave_at_two_waves_per_simd_full_chip(oracle, hiplib):
    """B = 6144: the default dispatch runs 2048 waves of three problems, two per SIMD (the instantiation no small batch reaches without
    forcing it).  Everything bit for bit as from the one-wave-per-SIMD build of the same mapping, and a sample against the CPU oracle."""
    from bunmpc_amd import batch as bbm
    B = 6144
    b = problems.make_batch("solo12_trot", B)
    out = {}
    for mode in (2, 0):
        old = hiplib.bmpc_set_two_waves_per_simd(mode)
        try:
            dev = bbm.DeviceBatch(b, num_iters=10, keep_hist=True)
            dev.solve()
            out[mode] = dev.results()
            assert hiplib.bmpc_biconvex_last_lanes_per_problem() == 21 and hiplib.bmpc_biconvex_last_waves_per_simd() == (2 if mode else 1)
        finally:
            hiplib.bmpc_set_two_waves_per_simd(old)
    for k in ("X", "F", "P", "L_x", "L_f", "stats", "trace", "hist", "dyn_viol"):
        assert np.array_equal(out[2][k], out[0][k], equal_nan=True), k
    idx = np.arange(0, B, B // 48)[:48]
    ref = oracle.solve_batch(b.take(idx), num_iters=10)
    assert np.array_equal(out[2]["stats"][idx], ref["stats"])
    for k in ("X", "F"):
        e = np.linalg.norm(out[2][k][idx] - ref[k], axis=1) / np.maximum(np.linalg.norm(ref[k], axis=1), 1e-300)
        assert e.max() < TOL, (k, e.max())


def test_work_stealing_kernel_at_two_waves_per_simd(hiplib):
    """The work-stealing kernel's two-waves-per-SIMD instantiation (never the default's choice: slower; a forced launch): a grid of 1536
    waves over 4099 problems at num_iters = 100, results bit for bit the one-wave launch's."""
    from bunmpc_amd import batch as bbm
    b = problems.make_batch("solo12_trot", 4099)
    out = {}
    for mode in (1, 0):
        old = hiplib.bmpc_set_two_waves_per_simd(mode)
        oldg = hiplib.bmpc_set_steal_grid(1536 if mode else 0)
        try:
            dev = bbm.DeviceBatch(b, num_iters=100, keep_hist=True)
            dev.solve()
            out[mode] = dev.results()
            assert hiplib.bmpc_biconvex_last_kernel_name().decode() == "biconvex_admm_steal_kernel" and hiplib.bmpc_biconvex_last_waves_per_simd() == mode + 1
        finally:
            hiplib.bmpc_set_two_waves_per_simd(old)
            hiplib.bmpc_set_steal_grid(oldg)
    for k in ("X", "F", "P", "L_x", "L_f", "stats", "trace"):
        assert np.array_equal(out[1][k], out[0][k]), k
    assert np.allclose(out[1]["dyn_viol"], out[0]["dyn_viol"], rtol=1e-12, atol=0)


def test_two_waves_per_simd_raw_form_is_bit_identical(hiplib, mapping):
    """... and the raw cost / bound form (set_cost_x / set_bounds_x / set_cost_f given explicitly, with and without a linear force
    cost), whose instantiations are separate kernels."""
    mapping("batch")
    b = problems.make_batch("solo12_trot", 11)
    rng = np.random.default_rng(5)
    nx, nf = 9 * (b.H + 1), 12 * b.H
    for with_qf in (False, True):
        raw = dict(Qx=rng.uniform(0.5, 50.0, (b.B, nx)), qx=rng.normal(0.0, 1.0, (b.B, nx)), lbx=np.full((b.B, nx), -2.0), ubx=np.full((b.B, nx), 2.0),
                   Qf=rng.uniform(1e-4, 1e-2, (b.B, nf)))
        if with_qf:
            raw["qf"] = rng.normal(0.0, 1e-3, (b.B, nf))
        out = {}
        for mode in (1, 0):
            old = hiplib.bmpc_set_two_waves_per_simd(mode)
            try:
                out[mode] = bb.solve_host(b, num_iters=4, keep_hist=True, raw=raw)
                assert hiplib.bmpc_biconvex_last_waves_per_simd() == (2 if mode else 1)
            finally:
                hiplib.bmpc_set_two_waves_per_simd(old)
        for k in ("X", "F", "P", "L_x", "L_f", "stats", "trace", "hist"):
            assert np.array_equal(out[1][k], out[0][k], equal_nan=True), (with_qf, k)


def test_work_stealing_does_not_change_results(hiplib):
    """num_iters = 100 (the reference's own call): the ADMM's early exit makes the iteration counts differ per problem, and the
    three-per-wave kernel runs as a persistent grid whose segments take the next unsolved problem when theirs has finished
    (bmpc_set_work_stealing).  Which segment solves a problem, and when, must not show in ANY output: everything bit for bit as
    from the plain three-per-wave launch -- iterates, step constants, counters, the per-iteration history.  B = 4099: more
    problems than the chip's segments (so segments do take second problems) and no multiple of three."""
    from bunmpc_amd import batch as bbm
    B = 4099
    b = problems.make_batch("solo12_trot", B)
    out = {}
    for on in (1, 0):
        old = hiplib.bmpc_set_work_stealing(on)
        try:
            dev = bbm.DeviceBatch(b, num_iters=100, keep_hist=True)
            dev.solve()
            out[on] = dev.results()
            name = hiplib.bmpc_biconvex_last_kernel_name().decode()
            assert name == ("biconvex_admm_steal_kernel" if on else "biconvex_admm_kernel") and hiplib.bmpc_biconvex_last_lanes_per_problem() == 21
            dev.solve()                      # a second launch: another counter of the ring, the same results
            again = dev.results()
            assert np.array_equal(again["X"], out[on]["X"]) and np.array_equal(again["stats"], out[on]["stats"])
        finally:
            hiplib.bmpc_set_work_stealing(old)
    n = out[1]["stats"][:, 0]
    print("work stealing: ADMM iterations min %d median %d max %d over %d problems" % (n.min(), np.median(n), n.max(), B))
    assert n.min() < 60 and n.max() == 100 and np.all(out[1]["stats"][:, 5] == 0)
    for k in ("X", "F", "P", "L_x", "L_f", "stats", "trace"):
        assert np.array_equal(out[1][k], out[0][k]), k
    # (the violation is a segment sum: its order of additions depends on WHICH of the wave's three segments holds the problem)
    assert np.allclose(out[1]["dyn_viol"], out[0]["dyn_viol"], rtol=1e-12, atol=0)
    assert np.allclose(out[1]["hist"], out[0]["hist"], rtol=1e-12, atol=0, equal_nan=True)
    # ... and from a WARM start with carried step constants (set_warm_start_vars; FISTA's L_ persists, fista.hpp:52): the segments
    # then read a problem's iterates and L from the arrays when they take it, not only at the start of the launch
    warm = {}
    for on in (1, 0):
        old = hiplib.bmpc_set_work_stealing(on)
        try:
            dev = bbm.DeviceBatch(b, num_iters=40)
            dev.set_warm_start(out[0]["X"], out[0]["F"], out[0]["P"], L_x=out[0]["L_x"] * 1.5, L_f=out[0]["L_f"] * 0.5)
            dev.solve()
            warm[on] = dev.results()
            assert hiplib.bmpc_biconvex_last_kernel_name().decode() == ("biconvex_admm_steal_kernel" if on else "biconvex_admm_kernel")
        finally:
            hiplib.bmpc_set_work_stealing(old)
    for k in ("X", "F", "P", "L_x", "L_f", "stats"):
        assert np.array_equal(warm[1][k], warm[0][k]), k
    assert len(np.unique(warm[1]["stats"][:, 0])) > 3


@pytest.mark.parametrize("which", ["batch", "wave"])
def test_early_exit_on_exit_tol(oracle, mapping, which):
    """||A_f X - b_f|| < exit_tol stops a problem (biconvex.cpp:111-114) while its wave-mate goes on."""
    mapping(which)
    b = problems.make_batch("solo12_trot", 2)
    ref = oracle.solve_batch(b, num_iters=12, exit_tol=0.06)
    got = bb.solve_host(b, num_iters=12, exit_tol=0.06)
    assert np.array_equal(got["stats"], ref["stats"]) and len(set(ref["stats"][:, 0])) > 1
    for k in "XFP":
        assert np.all(rel_l2(got[k], ref[k]) < TOL), k


def test_full_size_invariants_and_sampled_parity(oracle):
    """BASELINE size (4096 Solo12 trot problems, 10 ADMM iterations) through the device-resident
    path bench.py uses: size-independent properties on all problems, oracle parity on a sample."""
    from bunmpc_amd import batch as bbm
    B = 4096
    b = problems.make_batch("solo12_trot", B)
    dev = bbm.DeviceBatch(b, num_iters=10)
    dev.solve()
    got = dev.results()
    H, E = b.H, b.E
    assert np.all(got["stats"][:, 5] == 0) and np.all(got["stats"][:, 0] == 10)
    F = got["F"].reshape(B, H, E, 3)
    assert np.all(F[b.cnt_plan[..., 0] == 0] == 0.0)                  # swing feet carry no force
    s = F[..., 0] ** 2 + F[..., 1] ** 2
    assert np.all(F[..., 2] >= 0) and np.all(s <= F[..., 2] * (1 + 1e-9) + 1e-12)   # projection set
    X = got["X"].reshape(B, H + 1, 9)
    assert np.all(np.abs(X[:, 0] - b.x_init) < 5e-2)
    # P is the running sum of the violations: recompute the last violation independently
    sub = np.arange(0, B, 64)
    for i in sub[:8]:
        A, bf = oracle.dense_A_f(b.cnt_plan[i], b.dt[i], b.m, got["F"][i], b.x_init[i])
        assert abs(np.linalg.norm(A @ got["X"][i] - bf) - got["dyn_viol"][i]) < 1e-9
    # second solve from the same cold start is bit-identical (deterministic reductions)
    dev.solve()
    again = dev.results()
    for k in "XFP":
        assert np.array_equal(again[k], got[k])
    ref, spread = cpu_spread(b.take(sub), 10, oracle)
    err, bound = within_envelope({k: got[k][sub] for k in "XF"}, ref, spread)
    print("sampled parity over %d problems: median %.2e, max %.2e, above 1e-5: %d; CPU spread max %.2e"
          % (len(sub), np.median(err), err.max(), (err > TOL).sum(), spread.max()))
    assert np.median(err) < 1e-12
    assert np.all(err <= bound)         # 1e-5 wherever the CPU restatements agree (all 64 here: measured 7e-15)


@pytest.mark.parametrize("config,B,H", [("go2_bound", 256, 40), ("solo12_trot", 256, None)])
def test_fp32_variant_with_fp64_residual_check(oracle, config, B, H):
    """BASELINE config 3: fp32 iterates / operators / projections (precision = 1), every accept and
    exit decision and the dynamics violation reduced in fp64.  No fp32 reference exists: the fp32
    result is held to the fp64 kernel's (itself oracle-checked above) -- median 1e-4 rel-L2, and
    for problems in the chaotic regime (test_chaotic_envelope) the same envelope -- and the
    violation it reports is re-derived in fp64 numpy from the returned X, F."""
    b = problems.make_batch(config, B, H=H) if H else problems.make_batch(config, B)
    d64 = bb.solve_host(b, num_iters=10)
    d32 = bb.solve_host(b, num_iters=10, precision="f32")
    assert np.array_equal(d32["stats"][:, [0, 5]], d64["stats"][:, [0, 5]])      # same ADMM count, no NaN
    for k in ("X", "F"):
        err = rel_l2(d32[k], d64[k])
        print(config, k, "fp32 vs fp64 rel-L2: median %.2e, p99 %.2e, max %.2e" % (np.median(err), np.quantile(err, 0.99), err.max()))
        assert np.median(err) < 1e-4
        assert (err > 5e-3).mean() <= 0.1
    # fp32 iterates still satisfy the projection set exactly (the projection is the last thing applied)
    F = d32["F"].reshape(B, b.H, b.E, 3)
    assert np.all(F[b.cnt_plan[..., 0] == 0] == 0.0) and np.all(F[..., 2] >= 0)
    # residual check in fp64: ||A_f X - b_f|| from the returned iterates vs the kernel's own number
    for i in range(0, B, 32):
        A, bf = oracle.dense_A_f(b.cnt_plan[i], b.dt[i], b.m, d32["F"][i], b.x_init[i])
        r = np.linalg.norm(A @ d32["X"][i] - bf)
        assert abs(r - d32["dyn_viol"][i]) <= 1e-4 * max(r, 1e-3), (i, r, d32["dyn_viol"][i])
    # the raw cost form is fp64 only: refused with BMPC_BAD_ARG, nothing launched
    from bunmpc_amd import _lib
    nx, nf = 9 * (b.H + 1), 12 * b.H
    raw = dict(Qx=np.ones((B, nx)), qx=np.zeros((B, nx)), lbx=np.full((B, nx), -np.inf), ubx=np.full((B, nx), np.inf),
               Qf=np.ones((B, nf)))
    with pytest.raises(_lib.BmpcError) as e:
        bb.solve_host(b, num_iters=1, precision="f32", raw=raw)
    assert e.value.code == 1


@pytest.fixture
def mapping(hiplib):
    """selects the kernel: 'batch' = one knot per lane (biconvex_admm.hip), 'wave' = one problem per wave (biconvex_latency.hip)"""
    old = hiplib.bmpc_set_latency_mapping_max_batch(0)

    def choose(which):
        hiplib.bmpc_set_latency_mapping_max_batch({"batch": 0, "wave": 1 << 30}[which])
    yield choose
    hiplib.bmpc_set_latency_mapping_max_batch(old)


@pytest.mark.parametrize("config,B,H,iters", [("solo12_trot", 16, None, 10), ("solo12_trot_nominal", 1, None, 10), ("solo12_mixed", 12, None, 1),
                                              ("solo12_trot", 7, 3, 2), ("solo12_trot", 5, 15, 2), ("solo12_trot", 3, 19, 3)])
def test_latency_mapping_equals_batch_mapping_and_oracle(oracle, mapping, config, B, H, iters):
    """The one-problem-per-wave kernel (small batches, H <= 20) against the one-knot-per-lane kernel and the CPU oracle:
    identical discrete path (every iteration / retry count), values equal to rounding (its segment sums run over another
    lane order), 1e-5 to the oracle (measured ~1e-15)."""
    b = problems.make_batch(config, B, H=H) if H else problems.make_batch(config, B)
    ref = oracle.solve_batch(b, num_iters=iters)
    mapping("batch")
    kb = bb.solve_host(b, num_iters=iters, keep_hist=True)
    mapping("wave")
    kw = bb.solve_host(b, num_iters=iters, keep_hist=True)
    assert np.array_equal(kw["stats"], ref["stats"]) and np.array_equal(kw["stats"], kb["stats"])
    for k in "XFP":
        assert np.all(rel_l2(kw[k], ref[k]) < TOL), k
        assert np.all(rel_l2(kw[k], kb[k]) < 1e-12), k
    assert np.allclose(kw["hist"], kb["hist"], rtol=1e-10, equal_nan=True) and np.allclose(kw["dyn_viol"], kb["dyn_viol"], rtol=1e-10)
    assert np.array_equal(kw["L_x"], kb["L_x"]) and np.array_equal(kw["L_f"], kb["L_f"])


def test_fp32_shortcut_of_the_step_decisions_changes_nothing(hiplib, mapping):
    """The one-problem-per-wave kernel takes retry / exit decisions from fp32 wave sums when they are clear of their thresholds
    (biconvex_lanes.h::banded_decisions).  With the shortcut switched off every decision comes from the fp64 sums: the two runs
    must agree in EVERY BIT of every output -- full 10-iteration solves that pass through ~25 exits and a few thousand retry
    tests per problem, step constants low enough to force retries, and a diverging problem (inf / NaN sums)."""
    mapping("wave")
    b = problems.make_batch("solo12_mixed", 64)
    b.x_init[5, 2] = 1e200
    b.X_nom[5] = 1e200
    Lx = np.where(np.arange(64) % 3 == 0, 1e4, 2.25e6)
    Lf = np.where(np.arange(64) % 4 == 0, 10.0, 506.25)
    X0, F0, P0 = b.warm_start()

    def run(exact):
        old = hiplib.bmpc_set_exact_step_decisions(exact)
        try:
            return bb.solve_host(b, num_iters=10, warm=(X0, F0, P0), L_x=Lx, L_f=Lf, keep_hist=True)
        finally:
            hiplib.bmpc_set_exact_step_decisions(old)
    fast, exact = run(0), run(1)
    assert hiplib.bmpc_biconvex_last_kernel_name().decode() == "biconvex_latency_kernel"
    for k in ("X", "F", "P", "L_x", "L_f", "stats", "hist", "dyn_viol"):
        assert np.array_equal(fast[k], exact[k], equal_nan=True), k
    assert fast["stats"][:, 3].sum() > 0 and fast["stats"][:, 4].sum() > 0 and fast["stats"][5, 5] == 2


def test_latency_mapping_raw_form_warm_start_and_backtracking(oracle, mapping):
    """raw cost / bound arrays, warm start, per-problem L0 low enough to force retries in both FISTA loops, early exit, and a
    diverging problem -- all through the one-problem-per-wave kernel"""
    mapping("wave")
    b = problems.make_batch("solo12_trot", 6)
    pre = oracle.solve_batch(b, num_iters=0)
    raw = {k: pre[k] for k in ("Qx", "qx", "lbx", "ubx", "Qf")}
    raw["qf"] = 0.01 * np.random.default_rng(0).standard_normal(pre["Qf"].shape)
    Lx = np.array([2.25e6, 1e4, 1e5, 3e5, 2.25e6, 5e4])
    Lf = np.array([506.25, 10.0, 50.0, 506.25, 20.0, 100.0])
    X0, F0, P0 = b.warm_start()
    got = bb.solve_host(b, num_iters=3, raw=raw, warm=(X0, F0, P0), L_x=Lx, L_f=Lf)
    for i in range(b.B):
        r = oracle.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], pre["Qx"][i], pre["qx"][i], pre["Qf"][i], pre["lbx"][i],
                                  pre["ubx"][i], X0[i], F0[i], P0[i], L_x=Lx[i], L_f=Lf[i], rho=b.rho, num_iters=3, qf=raw["qf"][i])
        assert np.array_equal(got["stats"][i], r["stats"]), i
        assert got["L_x"][i] == r["L_x"] and got["L_f"][i] == r["L_f"]
        for k in "XFP":
            assert rel_l2(got[k][i], r[k]) < TOL, (i, k)
    assert got["stats"][:, 3].sum() > 0 and got["stats"][:, 4].sum() > 0
    # early exit on exit_tol, different ADMM counts inside one launch
    b2 = problems.make_batch("solo12_trot", 2)
    ref = oracle.solve_batch(b2, num_iters=12, exit_tol=0.06)
    g2 = bb.solve_host(b2, num_iters=12, exit_tol=0.06)
    assert np.array_equal(g2["stats"], ref["stats"]) and len(set(ref["stats"][:, 0])) > 1
    # NaN handling (biconvex.cpp:106-109)
    bad = problems.make_batch("solo12_trot", 3)
    bad.x_init[1, 2] = 1e200
    bad.X_nom[1] = 1e200
    g3 = bb.solve_host(bad, num_iters=4)
    assert g3["stats"][1, 5] == 2 and g3["stats"][1, 0] == 1 and not np.isfinite(g3["X"][1]).all()
    assert np.all(g3["stats"][[0, 2], 5] == 0) and np.isfinite(g3["X"][[0, 2]]).all()


def test_both_mappings_at_full_size(oracle, mapping):
    """B = 4096 through both kernels: same iteration statistics for every problem, values equal to rounding"""
    b = problems.make_batch("solo12_trot", 4096)
    out = {}
    for which in ("batch", "wave"):
        mapping(which)
        dev = bb.DeviceBatch(b, num_iters=10)
        dev.solve()
        out[which] = dev.results()
    assert np.array_equal(out["batch"]["stats"], out["wave"]["stats"])
    e = np.maximum(rel_l2(out["wave"]["X"], out["batch"]["X"]), rel_l2(out["wave"]["F"], out["batch"]["F"]))
    print("wave vs batch mapping over 4096 problems: median %.2e, max %.2e, above 1e-9: %d" % (np.median(e), e.max(), (e > 1e-9).sum()))
    assert np.median(e) < 1e-13 and (e > 1e-9).mean() < 0.02 and e.max() < 1e-3      # a few problems sit in the chaotic regime (tests/util.py)


def test_cold_start_that_carries_the_step_constants(oracle, mapping):
    """bmpc_batch_t.cold_start = 2: the next optimize call of the same KinoDynMP objects -- iterates reset by set_warm_starts
    (kino_dyn.cpp:83-99), FISTA's L_ carried from the call before (fista.hpp:52: it is set in the constructor only)."""
    b = problems.make_batch("solo12_trot", 5)
    pre = oracle.solve_batch(b, num_iters=0)
    X0, F0, P0 = b.warm_start()
    for which in ("batch", "wave"):
        mapping(which)
        dev = bb.DeviceBatch(b, num_iters=3)
        dev.cold_start(carry_step_constants=True)
        dev.set_step_constants(2e5, 40.0)            # low enough to force retries, which must persist into the second call
        Lx, Lf = np.full(5, 2e5), np.full(5, 40.0)
        for call in range(2):
            dev.solve()
            got = dev.results()
            for i in range(b.B):
                r = oracle.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], pre["Qx"][i], pre["qx"][i], pre["Qf"][i], pre["lbx"][i],
                                          pre["ubx"][i], X0[i], F0[i], P0[i], L_x=Lx[i], L_f=Lf[i], rho=b.rho, num_iters=3)
                assert np.array_equal(got["stats"][i], r["stats"]), (which, call, i)
                assert got["L_x"][i] == r["L_x"] and got["L_f"][i] == r["L_f"]
                assert rel_l2(got["X"][i], r["X"]) < TOL and rel_l2(got["F"][i], r["F"]) < TOL
                Lx[i], Lf[i] = r["L_x"], r["L_f"]
            if call == 0:
                assert got["stats"][:, 3:5].sum() > 0 and np.all(got["L_x"] > 2e5)
            else:
                assert got["stats"][:, 3:5].sum() == 0           # the constants found in the first call hold in the second
        dev.cold_start()                                         # a fresh object again: the constructor's constants
        dev.solve()
        assert np.all(dev.results()["L_x"] == 2.25e6)

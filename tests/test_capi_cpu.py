"""CPU tests of the C-ABI boundary (no GPU, no compute kernels): the library loads, exports
every symbol include/bunmpc.h declares, and its host-side pieces (gait planner, cost / bound
builders, debug matrices, argument checking) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from bunmpc_amd import _lib, problems
from bunmpc_amd.biconvex_mpc_cpp import BiconvexMP
from bunmpc_amd.gait_planner_cpp import GaitPlanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_list_the_same_symbols(hiplib):
    hdr = open(os.path.join(ROOT, "include", "bunmpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bmpc_[a-z_0-9A-Z]+)\s*\(", hdr))
    assert declared == set(_lib.exported_symbols())
    for name in declared:
        assert hasattr(hiplib, name), name          # dlsym of each declared entry point
    assert hiplib.bmpc_abi_version() == 2
    assert hiplib.bmpc_batch_struct_size() == C.sizeof(_lib.Batch)


def test_gait_planner_matches_oracle(oracle):
    lib = oracle.lib()
    rng = np.random.default_rng(11)
    for g in (problems.TROT, problems.BOUND, problems.JUMP):
        gp = GaitPlanner(g.gait_period, np.array(g.stance_percent), np.array(g.phase_offset), g.step_ht)
        for t in np.round(rng.uniform(0, 2.5, 50), 3):
            for j in range(4):
                sp, off = g.stance_percent[j], g.phase_offset[j]
                assert gp.get_phase(t, j) == lib.orc_gait_phase(t, g.gait_period, sp, off)
                assert gp.get_phi(t, j) == lib.orc_gait_phi(t, g.gait_period, off)
                assert gp.get_percent_in_phase(t, j) == lib.orc_gait_percent_in_phase(t, g.gait_period, sp, off)


def test_gait_planner_vector_overload_quirks():
    """gait_planner.cpp:31-39 writes only phi_[0]; :60-75 has no 1e-4 slack; both kept."""
    gp = GaitPlanner(0.5, np.array([0.6] * 4), np.array([0.0, 0.5, 0.5, 0.1]), 0.1)
    phi = gp.get_phi(0.07)
    assert phi[0] == pytest.approx(np.fmod(0.07 + 0.1 * 0.5, 0.5)) and np.all(phi[1:] == 0)
    assert gp.get_phase(0.30005, 0) == 1            # scalar overload: slack
    assert gp.get_phase(0.30005)[0] == 0            # vector overload: none
    plan = gp.get_contact_phase_plan(np.zeros((6, 4), dtype=int), 0.0, 0.05)
    assert plan.shape == (6, 4) and np.array_equal(plan[0], gp.get_phase(0.0))
    with pytest.raises(_lib.BmpcError):
        gp.get_phase(0.0, 7)


def _loaded_handle(b, i=0):
    mp = BiconvexMP(b.m, b.H, b.E)
    mp.set_rho(b.rho)
    for t in range(b.H):
        mp.set_contact_plan(b.cnt_plan[i, t], b.dt[i, t])
    return mp


def test_host_side_matrices_match_oracle(oracle):
    b = problems.make_batch("solo12_trot", 2)
    mp = _loaded_handle(b, 1)
    rng = np.random.default_rng(5)
    X = rng.standard_normal(9 * (b.H + 1))
    F = rng.standard_normal(3 * b.E * b.H)
    A, bx = oracle.dense_A_x(b.cnt_plan[1], b.dt[1], b.m, X)
    assert np.array_equal(mp.return_A_x(X), A) and np.array_equal(mp.return_b_x(X), bx)
    A, bf = oracle.dense_A_f(b.cnt_plan[1], b.dt[1], b.m, F, b.x_init[1])
    assert np.allclose(mp.return_A_f(F, b.x_init[1]), A, rtol=0, atol=1e-15)
    assert np.allclose(mp.return_b_f(F, b.x_init[1]), bf, rtol=0, atol=1e-15)


def test_argument_checking_and_append_semantics():
    b = problems.make_batch("solo12_trot_nominal", 1)
    mp = BiconvexMP(b.m, b.H, b.E)
    with pytest.raises(_lib.BmpcError):                     # plan incomplete
        mp.return_A_x(np.zeros(9 * (b.H + 1)))
    for t in range(b.H):
        mp.set_contact_plan(b.cnt_plan[0, t], b.dt[0, t])
    with pytest.raises(_lib.BmpcError):                     # H+1-th append (UB in the reference)
        mp.set_contact_plan(b.cnt_plan[0, 0], 0.05)
    with pytest.raises(_lib.BmpcError):                     # "bound constraints wrong size. Expected 6 ..."
        mp.create_bound_constraints(np.zeros((b.H, 5)), 15, 15, 15)
    with pytest.raises(ValueError):
        mp.create_cost_F(np.zeros(3))
    with pytest.raises(ValueError):                         # only diagonal Q
        mp.set_cost_x(np.ones((9 * (b.H + 1),) * 2), np.zeros(9 * (b.H + 1)))
    mp.set_cost_x(np.eye(9 * (b.H + 1)), np.zeros(9 * (b.H + 1)))
    assert mp.return_opt_com().shape == (b.H + 1, 3) and mp.return_opt_mom().shape == (b.H + 1, 6)
    assert mp.step_constants() == (2.25e6, 506.25)          # biconvex.cpp:20-21
    assert mp.return_dyn_viol_hist() == []


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly, not compute on the CPU, when no GPU is present."""
    n = C.c_int(0)
    rc = _lib.lib().bmpc_device_count(C.byref(n))
    if rc == _lib.OK and n.value > 0:
        pytest.skip("a GPU is present")
    b = problems.make_batch("solo12_trot_nominal", 1)
    mp = _loaded_handle(b)
    mp.create_bound_constraints(b.bounds[0], 15, 15, 15)
    mp.create_cost_X(b.W_X[0], b.W_X_ter[0], b.X_ter[0], b.X_nom[0])
    mp.create_cost_F(b.W_F[0])
    with pytest.raises(_lib.BmpcError) as e:
        mp.optimize(b.x_init[0], 2)
    assert e.value.code == _lib.DEVICE_ERROR
    from bunmpc_amd import batch
    with pytest.raises(_lib.BmpcError):
        batch.solve_host(b, num_iters=1)
    with pytest.raises(RuntimeError):
        batch.DeviceBatch(b)


def test_batch_descriptor_validation():
    d = _lib.Batch()
    _lib.lib().bmpc_batch_defaults(C.byref(d))
    assert (d.rho, d.mu, d.beta, d.tol, d.exit_tol, d.maxit) == (1e5, 1.0, 1.5, 1e-5, 1e-3, 150)
    d.B, d.n_col = 1, 20
    assert _lib.lib().bmpc_biconvex_solve_batch_host(C.byref(d)) == _lib.BAD_ARG      # missing arrays
    d.n_col = 256
    assert _lib.lib().bmpc_biconvex_solve_batch_host(C.byref(d)) == _lib.BAD_ARG
    assert b"256" in _lib.lib().bmpc_last_error()


def test_dispatch_knobs_round_trip(hiplib):
    """The dispatch switches of the batch entry points are plain process-wide settings: each setter returns the value it replaces
    (no GPU needed), and their defaults are the documented ones (include/bunmpc.h)."""
    for name, default, other in (("bmpc_set_three_per_wave", 2, 0), ("bmpc_set_two_waves_per_simd", 2, 1), ("bmpc_set_work_stealing", 1, 0),
                                 ("bmpc_set_steal_grid", 0, 1536), ("bmpc_set_latency_mapping_max_batch", 1024, 0)):
        f = getattr(hiplib, name)
        assert f(other) == default, name
        assert f(default) == other, name
    assert hiplib.bmpc_biconvex_last_waves_per_simd() in (1, 2)


def test_build_lock_serialises_builders(tmp_path):
    """the ranks of a multi-GPU job start together: whoever holds bunmpc_amd.build.build_lock() builds, the others wait"""
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); from bunmpc_amd import build\n"
            "with build.build_lock():\n"
            "    t0 = time.time(); time.sleep(0.5); print(t0, time.time())\n") % ROOT
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(3)]
    spans = sorted(tuple(float(v) for v in p.communicate()[0].split()) for p in ps)
    assert all(p.returncode == 0 for p in ps) and len(spans) == 3
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert b0 >= a1 - 1e-3           # no two holders at a time

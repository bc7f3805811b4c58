"""Per-problem CPU ensembles of the sampled problems of BASELINE's full-size configs.

    python tools/chaos_ensemble.py            # writes tests/golden/chaos_<config>.npz for the configs below

For every sampled problem: the strict C oracle, the matrix-free C variant, the numpy twin, and N_MEMBERS one-ulp perturbations
of x_init run through both C variants (tests/util.py::chaos_ensemble).  Stored per problem: k_calm (first ADMM iteration at which
the ensemble shows any sensitivity), the per-iteration relative range of the dynamics violation, the largest pairwise distance
of the final X / F, the range of the running counts, and the strict oracle's own history -- what the -m gpu tests hold the HIP
kernels to (tests/util.py::prefix_parity).  The `_c` fields come from the C members alone: tests/test_oracle_cpu.py reproduces
them on a subset in seconds; the fields without suffix add the numpy twin.  PARITY UNPINNED: every member is a restatement."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import problems          # noqa: E402
from oracle import oracle_c, oracle_np    # noqa: E402
from tests import util                    # noqa: E402

# (name, config, B, H, sample stride, ADMM iterations)
CASES = [("solo12_mixed", "solo12_mixed", 4096, None, 64, 10),
         ("go2_bound_h40", "go2_bound", 4096, 40, 64, 10),
         ("solo12_trot_n100", "solo12_trot", 4096, None, 64, 100),
         # the centroidal part of the reference's own call kd.optimize(q, v, 100, 1) (abstract_cyclic_gen.py:663) on the whole-body
         # batch of bench.py's KinoDyn legs: x_init = centroidal state of (q, v) by the CPU twin (kino_dyn.cpp:42)
         ("kinodyn_solo12_n100", "wb:solo12", 4096, None, 64, 100)]


def wb_dyn_batch(robot, B):
    """the centroidal batch inside problems.make_wb_batch(robot, B), x_init from the CPU twin's centroidal state"""
    from bunmpc_amd import urdf_model
    from oracle import ik_oracle_c as ic
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
    wb = problems.make_wb_batch(model, B)
    wb.dyn.x_init[:] = ic.centroidal_state(ic.Model(model), wb.x)
    return wb.dyn


def numpy_member(b, ref, iters):
    X0, F0, P0 = b.warm_start()
    out = dict(X=np.empty_like(ref["X"]), F=np.empty_like(ref["F"]), hist=np.full_like(ref["hist"], np.nan),
               trace=np.full_like(ref["trace"], -1))
    for i in range(b.B):
        r = oracle_np.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], ref["Qx"][i], ref["qx"][i], ref["Qf"][i], ref["lbx"][i],
                                     ref["ubx"][i], X0[i], F0[i], P0[i], rho=b.rho, num_iters=iters, mu=b.mu)
        n = len(r["hist"])
        out["X"][i], out["F"][i] = r["X"], r["F"]
        out["hist"][i, :n], out["trace"][i, :n] = r["hist"], r["trace"]
    return out


def make(name, config, B, H, stride, iters, with_numpy=True):
    if config.startswith("wb:"):
        b = wb_dyn_batch(config[3:], B)
    else:
        b = problems.make_batch(config, B, H=H) if H else problems.make_batch(config, B)
    sub = np.arange(0, B, stride)
    bs = b.take(sub)
    t0 = time.time()
    ref, ens_c = util.chaos_ensemble(bs, sub, iters, oracle_c)
    extra = [numpy_member(bs, ref, iters)] if with_numpy else []
    if extra:
        _, ens = util.chaos_ensemble(bs, sub, iters, oracle_c, extra=extra)
    else:
        ens = ens_c
    chaotic = ens["k_calm"] < iters
    print("%s: %d sampled problems, %d with a chaotic tail (k_calm %s), spread max %.2e, %.1f s"
          % (name, len(sub), chaotic.sum(), sorted(ens["k_calm"][chaotic].tolist()), ens["spread"].max(), time.time() - t0))
    for i in np.where(chaotic)[0]:
        print("  problem %4d: k_calm %2d  spread %.2e  violation range per iteration: %s" %
              (sub[i], ens["k_calm"][i], ens["spread"][i], " ".join("%.0e%s" % (x, "*" if c else "") for x, c in zip(ens["hist_spread"][i], ens["count_range"][i]))))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chaos_%s.npz" % name), sub=sub, iters=iters, members=util.N_MEMBERS,
                        k_calm=ens["k_calm"], hist_spread=ens["hist_spread"], spread=ens["spread"], count_range=ens["count_range"],
                        k_calm_c=ens_c["k_calm"], hist_spread_c=ens_c["hist_spread"], spread_c=ens_c["spread"],
                        ref_hist=ref["hist"], ref_trace=ref["trace"], ref_stats=ref["stats"])


if __name__ == "__main__":
    only = sys.argv[1:]
    for case in CASES:
        if not only or case[0] in only:
            make(*case)

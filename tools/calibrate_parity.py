#!/usr/bin/env python3
"""Prints the measurements the tolerances of tests/test_parity_envelope_gpu.py are derived from (run on the GPU box):
per problem, the distance of the GPU result to the strict C restatement next to the spread of the CPU restatements among
themselves (strict C, numpy twin, matrix-free C) on the same problems."""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import batch as bb, problems, urdf_model      # noqa: E402
from oracle import oracle_c, oracle_np                         # noqa: E402
from tests.util import rel_l2                                  # noqa: E402


def np_solve(b, i, ref, iters):
    X0, F0, P0 = b.warm_start()
    return oracle_np.biconvex_solve(b.cnt_plan[i], b.dt[i], b.m, b.x_init[i], ref["Qx"][i], ref["qx"][i], ref["Qf"][i], ref["lbx"][i],
                                    ref["ubx"][i], X0[i], F0[i], P0[i], rho=b.rho, num_iters=iters, mu=b.mu)


def cpu_spread(b, iters, with_np=True):
    ref = oracle_c.solve_batch(b, num_iters=iters)
    fast = oracle_c.solve_batch(b, num_iters=iters, fast=True)
    s = np.maximum(rel_l2(fast["X"], ref["X"]), rel_l2(fast["F"], ref["F"]))
    if with_np:
        for i in range(b.B):
            rn = np_solve(b, i, ref, iters)
            s[i] = max(s[i], rel_l2(rn["X"], ref["X"][i]), rel_l2(rn["F"], ref["F"][i]), rel_l2(rn["X"], fast["X"][i]), rel_l2(rn["F"], fast["F"][i]))
    return ref, s


def report(tag, b, iters, precision="f64", with_np=True):
    t0 = time.time()
    ref, s = cpu_spread(b, iters, with_np)
    got = bb.solve_host(b, num_iters=iters, precision=precision)
    e = np.maximum(rel_l2(got["X"], ref["X"]), rel_l2(got["F"], ref["F"]))
    ratio = e / np.maximum(s, 1e-300)
    chaotic = s > 1e-9
    print("%-40s B=%d iters=%d %s  (%.0f s)" % (tag, b.B, iters, precision, time.time() - t0))
    print("   admm counts equal: %s   status equal: %s" % (np.array_equal(got["stats"][:, 0], ref["stats"][:, 0]), np.array_equal(got["stats"][:, 5], ref["stats"][:, 5])))
    print("   calm problems (cpu spread <= 1e-9): %d, gpu err max %.2e median %.2e" % ((~chaotic).sum(), e[~chaotic].max() if (~chaotic).any() else 0, np.median(e[~chaotic]) if (~chaotic).any() else 0))
    if chaotic.any():
        print("   chaotic problems: %d, cpu spread median %.2e max %.2e | gpu err median %.2e max %.2e | ratio median %.2f max %.2f"
              % (chaotic.sum(), np.median(s[chaotic]), s[chaotic].max(), np.median(e[chaotic]), e[chaotic].max(), np.median(ratio[chaotic]), ratio[chaotic].max()))
    sys.stdout.flush()
    return e, s


def main():
    for cfg, B in (("solo12_mixed", 12), ("go2_bound", 6)):
        report("chaotic_envelope " + cfg, problems.make_batch(cfg, B), 10)
    report("hundred_admm solo12_trot", problems.make_batch("solo12_trot", 5), 100)
    for H in (40, 63):
        report("horizons solo12_trot H=%d" % H, problems.make_batch("solo12_trot", 7, H=H), 2 if H < 40 else 1)
    # full size samples
    for cfg, H in (("solo12_trot", None), ("go2_bound", 40), ("solo12_mixed", None)):
        b = problems.make_batch(cfg, 4096, H=H) if H else problems.make_batch(cfg, 4096)
        sub = np.arange(0, 4096, 64)
        report("full-size sample " + cfg, b.take(sub), 10)
    # fp32 against the oracle
    for cfg, B, H in (("go2_bound", 64, 40), ("solo12_trot", 64, None)):
        b = problems.make_batch(cfg, B, H=H) if H else problems.make_batch(cfg, B)
        e, s = report("fp32 vs oracle " + cfg, b, 10, precision="f32")
        calm = s <= 1e-9
        print("   fp32 calm problems: err quantiles 50/90/99/100 %%: %s" % np.quantile(e[calm], [0.5, 0.9, 0.99, 1.0]))
        print("   fp32 all problems:  err quantiles 50/90/99/100 %%: %s" % np.quantile(e, [0.5, 0.9, 0.99, 1.0]))
        e64, _ = report("fp64 same problems " + cfg, b, 10, with_np=False)
    # projection-set property for mu = 10
    b = problems.make_batch("go2_bound", 256, H=40)
    got = bb.solve_host(b, num_iters=10)
    F = got["F"].reshape(b.B, b.H, b.E, 3)
    sq = F[..., 0] ** 2 + F[..., 1] ** 2
    print("go2 mu=%g: max(s - mu fz) = %.3e, min fz %.3e, swing forces zero: %s" % (b.mu, (sq - b.mu * F[..., 2]).max(), F[..., 2].min(), np.all(F[b.cnt_plan[..., 0] == 0] == 0.0)))

    # Go2 H=60 DDP: problems that hit maxiter, GPU vs the C twin
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    from oracle import ik_oracle_c as ic
    go2 = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "go2.json")).read())
    wb = problems.make_wb_batch(go2, 16, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    kb = KinoDynDeviceBatch(wb, go2, num_iters=10)
    kb.solve()
    g = kb.results()
    m = ic.Model(go2)
    r = ic.solve_wb_batch(m, wb, g["X"], trace=True)
    print("go2 h60 gpu iters", g["ik_iters"], "status", g["ik_status"])
    print("go2 h60 twin iters", r["iters"], "status", r["status"])
    for i in range(16):
        n = int(min(g["ik_iters"][i], r["iters"][i]))
        tg, tc = g["ik_trace"][i, :n], r["trace"][i, :n]
        same_reg = np.array_equal(tg[:, 1], tc[:, 1])
        same_alpha = np.array_equal(tg[:, 2], tc[:, 2])
        crel = np.abs(tg[:, 0] - tc[:, 0]) / np.abs(tc[:, 0])
        print("  %2d iters %3d/%3d reg-seq equal %s alpha-seq equal %s cost rel max %.2e final cost rel %.2e xs rel %.2e stop rel max %.2e"
              % (i, g["ik_iters"][i], r["iters"][i], same_reg, same_alpha, crel.max(), abs(g["ik_cost"][i] - r["cost"][i]) / abs(r["cost"][i]),
                 rel_l2(g["xs"][i].reshape(-1), r["xs"][i].reshape(-1)), (np.abs(tg[:, 3] - tc[:, 3]) / np.abs(tc[:, 3])).max()))
        if not same_alpha:
            k = int(np.argmax(tg[:, 2] != tc[:, 2]))
            print("       first alpha difference at iteration", k, tg[k], tc[k])


if __name__ == "__main__":
    main()

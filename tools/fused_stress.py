#!/usr/bin/env python3
"""Repeated solves with the express lane / the forced lane / the fused-direct path against the multi-kernel path: every result
bit for bit, many times over (a race in the fused kernel's tick protocol shows as a rare mismatch).  usage: tools/fused_stress.py [reps]"""
import dataclasses, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import _lib, problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = _lib.lib()
robots = os.path.join(ROOT, "bunmpc_amd", "robots")
solo = urdf_model.RobotModel.from_json(open(os.path.join(robots, "solo12.json")).read())
go2 = urdf_model.RobotModel.from_json(open(os.path.join(robots, "go2.json")).read())
KEYS = ("xs", "us", "ik_cost", "ik_stop", "ik_iters", "ik_status")
bad = 0


def solve(model, wb, sched, fused_direct=16):
    old = lib.bmpc_ik_set_fused_direct_max(fused_direct)
    try:
        kb = KinoDynDeviceBatch(wb, model, num_iters=10, schedule=sched)
        kb.solve()
        return kb.results()
    finally:
        lib.bmpc_ik_set_fused_direct_max(old)


cases = []
for B in (4096, 1024, 640):
    cases.append(("solo12 B=%d express" % B, solo, problems.make_wb_batch(solo, B, seed=100 + B), {"express_cap": 96}))
cases.append(("solo12 B=12 fused-direct", solo, problems.make_wb_batch(solo, 12, seed=7), {}))
g = problems.make_wb_batch(go2, 160, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
cases.append(("go2 B=160 forced lane", go2, g, {"express_cap": 20, "debug_inject": 2}))
for name, model, wb, sched in cases:
    ref = solve(model, wb, {"express_cap": -1}, fused_direct=0)
    for r in range(reps):
        got = solve(model, wb, sched)
        diff = [k for k in KEYS if not np.array_equal(got[k], ref[k])]
        took = int((got["ik_fused_iters"] > 0).sum())
        if diff or took == 0:
            bad += 1
            print("MISMATCH", name, "rep", r, diff, "fused problems", took)
    print(name, "ok x%d (fused problems per solve: %d)" % (reps, took))
print("fused stress:", "FAILED %d" % bad if bad else "all identical")
sys.exit(1 if bad else 0)

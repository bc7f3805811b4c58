#!/usr/bin/env python3
"""which problems differ between a solve with the express lane and one without (debugging aid)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "solo12.json")).read())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wb = problems.make_wb_batch(model, B)
out = {}
for name, cap in (("off", -1), ("on", 96), ("on2", 96)):
    kb = KinoDynDeviceBatch(wb, model, num_iters=10, schedule={"express_cap": cap})
    kb.solve()
    out[name] = kb.results()
for name in ("on", "on2"):
    a, b = out["off"], out[name]
    bad = np.where(np.any(a["xs"].reshape(B, -1) != b["xs"].reshape(B, -1), axis=1) | (a["ik_iters"] != b["ik_iters"]))[0]
    print(name, "problems that differ:", bad.tolist())
    for i in bad[:6]:
        n = max(a["ik_iters"][i], b["ik_iters"][i])
        print("  problem", i, "iters off/on", a["ik_iters"][i], b["ik_iters"][i], "status", a["ik_status"][i], b["ik_status"][i], "fused iters", b["ik_fused_iters"][i])
        ta, tb = a["ik_trace"][i, :n], b["ik_trace"][i, :n]
        d = np.where(np.any(ta != tb, axis=1))[0]
        print("   first differing iteration", d[:3].tolist(), "off", ta[d[0]].tolist() if len(d) else None, "on", tb[d[0]].tolist() if len(d) else None)

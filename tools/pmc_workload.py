#!/usr/bin/env python3
"""one warm-up + N timed batch solves of a bench workload, nothing else (the program rocprofv3's counter passes run).
usage: tools/pmc_workload.py biconvex|go2_bound_f32|go2_bound_f64|solo12_h20|solo12_n100|go2_h60 [N=2]"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bunmpc_amd import batch as bb, problems, urdf_model
what, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
if what == "biconvex":
    job = bb.DeviceBatch(problems.make_batch("solo12_trot", 4096), num_iters=10).solve
elif what.startswith("go2_bound"):      # BASELINE config 3's workload
    job = bb.DeviceBatch(problems.make_batch("go2_bound", 4096), num_iters=10, precision="f32" if what.endswith("f32") else "f64").solve
else:
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    robot = "go2" if what == "go2_h60" else "solo12"
    iters = 100 if what == "solo12_n100" else 10      # solo12_n100: the reference's own call kd.optimize(q, v, 100, 1)
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
    if what == "go2_h60":
        wb = problems.make_wb_batch(model, 1024, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    else:
        wb = problems.make_wb_batch(model, 4096)
    job = KinoDynDeviceBatch(wb, model, num_iters=iters).solve
for _ in range(N + 1):
    job()
torch.cuda.synchronize()
print("solves", N + 1)

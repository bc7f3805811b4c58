#!/usr/bin/env python3
"""ONE batch of KinoDynMP.optimize problems split into k cohorts solved concurrently on k HIP streams (bunmpc_amd/pipeline.py):
wall time of the whole batch against k = 1.  usage: tools/cohort_probe.py solo12_h20|go2_h60 [B]"""
import dataclasses, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bunmpc_amd import problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
from bunmpc_amd.pipeline import StreamPool
cfg = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if cfg == "go2_h60" else 4096)
robot = "go2" if cfg == "go2_h60" else "solo12"
model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
if cfg == "go2_h60":
    wb = problems.make_wb_batch(model, B, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
else:
    wb = problems.make_wb_batch(model, B)
pools = {k: StreamPool(n_streams=k) for k in (2, 3)}
for k in (1, 2, 3, 4, 6):
    idx = np.array_split(np.arange(B), k)
    kbs = [KinoDynDeviceBatch(wb.take(i), model, num_iters=10) for i in idx]
    pool = pools[min(k, 3)] if k > 1 else None
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pool is None:
            kbs[0].solve()
        else:
            pool.run([kb.solve for kb in kbs])
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%s B=%d in %d cohorts: %.2f ms" % (cfg, B, k, best * 1e3), flush=True)

#!/usr/bin/env python3
"""Solo12 KinoDyn batch (B = 4096) over the express lane's capacity: ms per batch solve"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bunmpc_amd import _lib, problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
lib = _lib.lib()
model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "solo12.json")).read())
wb = problems.make_wb_batch(model, 4096)
for cap in (96, 0, 48, 64, 80, 112, 128, 160, 192, 96):
    old = lib.bmpc_ik_set_express_capacity(cap)
    kd = KinoDynDeviceBatch(wb, model, num_iters=10)
    kd.solve(); kd.solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        kd.solve()
    torch.cuda.synchronize()
    print("express capacity %3d: %.2f ms per batch solve" % (cap, (time.perf_counter() - t0) / 6 * 1e3), flush=True)
    lib.bmpc_ik_set_express_capacity(old)

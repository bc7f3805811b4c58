"""Batch-solve time against the active-problem threshold below which the forward pass uses its speculative mappings
(bmpc_ik_set_speculative_below; results do not depend on it).  usage (GPU box): python tools/spec_threshold_probe.py"""
import dataclasses
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bunmpc_amd import _lib, problems, urdf_model  # noqa: E402
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for robot, B, kw in (("solo12", 4096, {}), ("go2", 1024, dict(gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB))):
    model = urdf_model.RobotModel.from_json(open(os.path.join(root, "bunmpc_amd", "robots", robot + ".json")).read())
    wb = problems.make_wb_batch(model, B, **kw)
    kb = KinoDynDeviceBatch(wb, model, device="cuda:0")
    kb.solve()
    for below in (256, 512, 1024, 1536, 2048, 4096):
        old = _lib.lib().bmpc_ik_set_speculative_below(below)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            kb.solve()
        torch.cuda.synchronize()
        print(robot, "B", B, "speculative below", below, "ms per batch %.2f" % ((time.perf_counter() - t) / 3 * 1e3))
        _lib.lib().bmpc_ik_set_speculative_below(old)

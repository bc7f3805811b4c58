#!/usr/bin/env python3
"""Go2 / Solo12 KinoDyn batch with the one-wave speculative line search above N active problems: ms per solve and a digest of the results"""
import dataclasses, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bunmpc_amd import _lib, problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
lib = _lib.lib()
for robot in ("go2", "solo12"):
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
    if robot == "go2":
        wb = problems.make_wb_batch(model, 1024, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    else:
        wb = problems.make_wb_batch(model, 4096)
    for above in (0, 512, 768, 0, 512):
        lib.bmpc_ik_set_spec_one_wave_above(above)
        kd = KinoDynDeviceBatch(wb, model, num_iters=10)
        kd.solve(); kd.solve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            kd.solve()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4 * 1e3
        r = kd.results()
        h = hashlib.sha1()
        for k in sorted(r):
            if isinstance(r[k], np.ndarray): h.update(np.ascontiguousarray(r[k]).tobytes())
        print("%-7s one wave above %4d: %.2f ms per batch solve, digest %s" % (robot, above, dt, h.hexdigest()[:16]), flush=True)
lib.bmpc_ik_set_spec_one_wave_above(0)

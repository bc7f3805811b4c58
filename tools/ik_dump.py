#!/usr/bin/env python3
"""solve a KinoDyn batch and dump the results (for bit-comparisons between library builds: BUNMPC_LIB=...)
usage: tools/ik_dump.py solo12_h20|go2_h60 B out.npz"""
import dataclasses, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
cfg, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
robot = "go2" if cfg == "go2_h60" else "solo12"
model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
wb = problems.make_wb_batch(model, B, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB) if cfg == "go2_h60" else problems.make_wb_batch(model, B)
kb = KinoDynDeviceBatch(wb, model, num_iters=10)
kb.solve()
r = kb.results()
np.savez(out, xs=r["xs"], us=r["us"], iters=r["ik_iters"], cost=r["ik_cost"], X=r["X"])
print("iters mean %.3f max %d" % (r["ik_iters"].mean(), r["ik_iters"].max()))

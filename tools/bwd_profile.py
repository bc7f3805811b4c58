"""Cycle buckets of ik_backward_kernel (build with -DBWD_PROFILE): per-node average of
[stage inputs, apply+transpose+apply, Cholesky, gain solve, Schur+symmetrise] for a lone problem and in a full batch."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from bunmpc_amd import problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
from bunmpc_amd import _lib
import dataclasses, os
if os.environ.get("IK_FUSED_DIRECT"):       # 0: the lone problem through the lock-step kernels (two-wave Riccati kernel), not the fused one
    _lib.lib().bmpc_ik_set_fused_direct_max(int(os.environ["IK_FUSED_DIRECT"]))
robot = sys.argv[1] if len(sys.argv) > 1 else "solo12"      # solo12 (H = 20, H_ik = 10) | go2 (H = 60, H_ik = 30)
model = urdf_model.RobotModel.from_json(open("bunmpc_amd/robots/%s.json" % robot).read())
for B in ((1, 4096) if robot == "solo12" else (1, 1024)):
    if robot == "go2":
        wb = problems.make_wb_batch(model, B, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    else:
        wb = problems.make_wb_batch(model, B)
    kb = KinoDynDeviceBatch(wb, model, num_iters=10)
    kb.solve(); r = kb.results()
    T = wb.ik_T
    oq = kb.off["k"] + T * 18      # the Qu slot (unused by the solver: the stamped build parks its cycle counts there)
    o = oq + T * 18                # the Quuk slot
    ws = kb.ws.cpu().numpy()
    c = ws[:, o:o + 9] / wb.ik_T      # last backward pass only
    names = "stage apply1 row+apply2 qxu+Lxx chol solve store+vx schur sym+fs".split()
    cd = ws[:, oq:oq + 4].mean(0)
    fw = ws[:, oq + 8:oq + 14].mean(0)
    print("B", B, "forward (last call) cycles to the node barrier / incl. barrier, summed over nodes and rounds: chain %d / %d, legs %d / %d, reg %d / %d" % tuple(fw))
    ph = ws[:, oq + 16:oq + 21].mean(0)
    print("B", B, "forward chain wave, cycles summed over nodes and rounds: dx/residual %d, (legs on one-wave builds) %d, feedback %d, Euler %d, barriers + bookkeeping %d" % tuple(ph))
    print("B", B, "forward, the sum of a node's parts on its wave, cycles summed over nodes and rounds: %d" % ws[:, oq + 22].mean())
    print("B", B, "calcdiff node 0 cycles: stage %d, walk (|| fetch of the ik_state_kernel terms) %d, totals+columns %d, assembly %d" % tuple(cd))
    print("B", B, "cycles/node:", " ".join("%s %d" % (n, v) for n, v in zip(names, c.mean(0))), "| sum", round(c.mean(0).sum()))

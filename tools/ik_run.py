#!/usr/bin/env python3
"""two batch solves of KinoDynMP.optimize (the second is the one to look at in a kernel trace), printing the wall time.
usage: tools/ik_run.py solo12_h20|go2_h60 [B] [spec_below] [all_steps_below] [gains_wave_below]"""
import dataclasses, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bunmpc_amd import _lib, problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
cfg = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if cfg == "go2_h60" else 4096)
lib = _lib.lib()
if len(sys.argv) > 3:
    lib.bmpc_ik_set_speculative_below(int(sys.argv[3]))
if len(sys.argv) > 4:
    lib.bmpc_ik_set_all_steps(int(sys.argv[4]))
if len(sys.argv) > 5:
    lib.bmpc_ik_set_gains_wave_below(int(sys.argv[5]))
if os.environ.get("IK_EXPRESS"):
    lib.bmpc_ik_set_express_capacity(int(os.environ["IK_EXPRESS"]))
if os.environ.get("IK_EXPRESS_NEAR"):
    lib.bmpc_ik_set_express_near(float(os.environ["IK_EXPRESS_NEAR"]))
if os.environ.get("IK_FUSED_DIRECT"):
    lib.bmpc_ik_set_fused_direct_max(int(os.environ["IK_FUSED_DIRECT"]))
if os.environ.get("IK_BLOCKING_WAITS"):
    lib.bmpc_ik_set_blocking_waits(int(os.environ["IK_BLOCKING_WAITS"]))
robot = "go2" if cfg == "go2_h60" else "solo12"
model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
if cfg == "go2_h60":
    wb = problems.make_wb_batch(model, B, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
else:
    wb = problems.make_wb_batch(model, B)
kb = KinoDynDeviceBatch(wb, model, num_iters=10)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kb.solve()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    kb.solve_ik_only()
    torch.cuda.synchronize()
    print("solve %.2f ms, ik only %.2f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
r = kb.results()
print("iters mean %.2f max %d not converged %d" % (r["ik_iters"].mean(), r["ik_iters"].max(), (r["ik_status"] != 0).sum()))
# per-kernel split of one IK batch solve (events around every kernel of the DDP loop) and a digest of the results (to compare builds)
import ctypes, hashlib
lib.bmpc_ik_set_profile(1)
kb.solve_ik_only()
torch.cuda.synchronize()
lib.bmpc_ik_set_profile(0)
prof = (5 * ctypes.c_double)()
lib.bmpc_ik_last_profile(prof)
print("kernel ms per batch solve: state %.2f calcdiff %.2f backward %.2f forward %.2f other %.2f" % tuple(prof))
h = hashlib.sha256()
for k in ("xs", "us", "ik_iters", "ik_cost"):
    h.update(np.ascontiguousarray(r[k]).tobytes())
print("results digest", h.hexdigest()[:16], "ddp loop iterations", r["ddp_loop_iters"])
# the fused kernel's telemetry of the longest-running problem (cycles in derivative + Riccati pass / line search, ticks, iterations)
i = int(np.argmax(r["ik_iters"]))
oq = kb.off["k"] + wb.ik_T * 18
t = kb.ws[i, oq:oq + 12].cpu().numpy()
if t[3] > 0 and t[3] < 200:
    print("fused kernel, problem %d (%d iterations): %d turns, cycles per turn: phase A %.0f (%.1f ticks), phase B %.0f" % (i, r["ik_iters"][i], t[3], t[0] / t[3], t[2] / t[3], t[1] / t[3]))
    print("   recursion wave per turn: start -> first node %.0f, nodes %.0f, after the last node %.0f; line search inside its role %.0f" % tuple(t[4:8] / t[3]))
    print("   cycles per turn at the tick barriers: recursion %.0f, gains %.0f, producers %.0f / %.0f" % tuple(t[8:12] / t[3]))
al = kb.active_list.cpu().numpy()
print("express lane: xmeta (taken, count, iteration, -)", al[-260:-256].tolist(), "near counts", al[-262:-260].tolist())

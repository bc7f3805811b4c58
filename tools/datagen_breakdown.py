"""Where a PlanLabelGenerator.step goes (wall time per stage, synchronised): usage (GPU box): python tools/datagen_breakdown.py [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bunmpc_amd import datagen, problems, urdf_model  # noqa: E402
from bunmpc_amd import mpc_batch, perturbation, plan_batch, robot_id_controller  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = urdf_model.RobotModel.from_json(open(os.path.join(root, "bunmpc_amd", "robots", "solo12.json")).read())
wb = problems.make_wb_batch(model, B)
dev = "cuda:0"
gen = datagen.PlanLabelGenerator(model, device=dev)
up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
q, v = up(np.tile(problems.SOLO12_Q0, (B, 1))), torch.zeros((B, 18), dtype=torch.float64, device=dev)
t0, vdes = up(wb.dyn.meta["t0"]), up(wb.dyn.meta["v_des_body"])
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return wrapper


perturbation.PerturbationSampler.sample = timed("sampler.sample", perturbation.PerturbationSampler.sample)
plan_batch.DeviceWbPlan.build = timed("wb plan build", plan_batch.DeviceWbPlan.build)
plan_batch.DeviceWbPlan.update = timed("wb plan update", plan_batch.DeviceWbPlan.update)
mpc_batch.interpolate_on_device = timed("interpolate x3", mpc_batch.interpolate_on_device)
datagen.id_batch_device = timed("id rows", datagen.id_batch_device)
from bunmpc_amd import kinodyn_batch  # noqa: E402
kinodyn_batch.KinoDynDeviceBatch.solve = timed("kinodyn solve", kinodyn_batch.KinoDynDeviceBatch.solve)
for it in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t = time.perf_counter()
    gen.step(q, v, t0, vdes)
    torch.cuda.synchronize()
    total = time.perf_counter() - t
print("total %.2f ms" % (total * 1e3), {k: round(x * 1e3, 2) for k, x in acc.items()}, "other %.2f ms" % ((total - sum(acc.values())) * 1e3))

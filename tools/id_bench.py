"""Timing of the inverse-dynamics output stage (bmpc_id_batch_device): n samples = B plans x rows of 1 kHz plan.
usage (GPU box): python tools/id_bench.py [B] [rows]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bunmpc_amd import robot_id_controller as ric, urdf_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n = B * rows
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = urdf_model.RobotModel.from_json(open(os.path.join(root, "bunmpc_amd", "robots", "solo12.json")).read())
ctrl = ric.InverseDynamicsController(model, ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"])
ctrl.set_gains(3.0, 0.05)
g = torch.Generator(device="cuda:0").manual_seed(1)
xs = torch.randn((n, 37), dtype=torch.float64, device="cuda:0", generator=g) * 0.3
xs[:, 3:7] = torch.nn.functional.normalize(torch.randn((n, 4), dtype=torch.float64, device="cuda:0", generator=g), dim=1)
us = torch.randn((n, 18), dtype=torch.float64, device="cuda:0", generator=g)
f = torch.randn((n, 12), dtype=torch.float64, device="cuda:0", generator=g) * 5
q = xs[:, :19] + 0.01
v = xs[:, 19:] + 0.01
for want in (("tau_ff", "tau_fb", "action", "state"), ("action",)):
    for _ in range(3):
        ctrl.rows(xs, us, f, q, v, want=want)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ctrl.rows(xs, us, f, q, v, want=want)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bytes_in = (37 + 18 + 12 + 19 + 18) * 8
    bytes_out = sum(43 if w == "state" else 12 for w in want) * 8
    print(json.dumps(dict(n=n, want=list(want), ms=ms, samples_per_s=n / ms * 1e3, algorithmic_GBps=n * (bytes_in + bytes_out) / ms / 1e6,
                          frac_of_8TBps=n * (bytes_in + bytes_out) / ms / 1e6 / 8000)))

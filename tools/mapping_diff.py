"""lock-step mappings against each other on a large batch (no express lane): line search always four problems per wave vs the default
thresholds -- which problems differ, and by how much (debugging aid for rare last-bit differences between the two robot walks)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import _lib, problems, urdf_model
from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
lib = _lib.lib()
solo = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "solo12.json")).read())
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4196
wb = problems.make_wb_batch(solo, 4096, seed=seed)
out = {}
for name, below in (("default", 1024), ("never_spec", 0), ("always_spec", 1 << 30)):
    old = lib.bmpc_ik_set_speculative_below(below)
    kb = KinoDynDeviceBatch(wb, solo, num_iters=10, schedule={"express_cap": -1})
    kb.solve(); out[name] = kb.results()
    lib.bmpc_ik_set_speculative_below(old)
for name in ("never_spec", "always_spec"):
    a, b = out["default"], out[name]
    for k in ("ik_cost", "xs", "us", "ik_iters"):
        d = np.where(np.any((a[k] != b[k]).reshape(4096, -1), axis=1))[0]
        print(name, k, "differs for", d.tolist()[:10], len(d))

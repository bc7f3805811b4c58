"""Does pairing similar problems in a wave (LPP = 32: two problems per wave run each FISTA loop until both are done)
shorten the launch?  Same 4096 problems, different order."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from bunmpc_amd import problems, batch as bb
b = problems.make_batch("solo12_trot", 4096)
def run(bt, label):
    dev = bb.DeviceBatch(bt, num_iters=10)
    dev.solve(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dev.solve(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    r = dev.results()
    print("%-28s %.3f ms   fista iters/solve %.0f" % (label, min(ts), r["stats"][:, 1:3].sum(1).mean()))
    return r
r0 = run(b, "as generated")
t0 = b.meta["t0"]
run(b.take(np.argsort(t0, kind="stable")), "sorted by start phase t0")
its = r0["stats"][:, 1:3].sum(1)
run(b.take(np.argsort(its, kind="stable")), "sorted by total iterations (oracle knowledge)")
key = t0 * 1000 + b.meta["v_des"][:, 0]
run(b.take(np.argsort(key, kind="stable")), "sorted by (t0, v_des)")

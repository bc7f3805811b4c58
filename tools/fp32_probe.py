import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from bunmpc_amd import problems, batch as bb
def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
for cfg, B, H in (("solo12_trot", 4096, None), ("go2_bound", 4096, 40), ("solo12_mixed", 4096, None)):
    b = problems.make_batch(cfg, B, H=H) if H else problems.make_batch(cfg, B)
    out = {}
    for prec in ("f64", "f32"):
        dev = bb.DeviceBatch(b, num_iters=10, precision=prec)
        dev.solve(); torch.cuda.synchronize()
        t = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dev.solve(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
        out[prec] = dev.results(); out[prec]["ms"] = min(t)
    a, c = out["f32"], out["f64"]
    ex, ef = rel(a["X"], c["X"]), rel(a["F"], c["F"])
    print(cfg, "H", b.H, "ms f64 %.2f f32 %.2f" % (c["ms"], a["ms"]))
    print("  relX med %.2e max %.2e  relF med %.2e max %.2e" % (np.median(ex), ex.max(), np.median(ef), ef.max()))
    print("  admm iters f64", np.bincount(c["stats"][:, 0]), "f32", np.bincount(a["stats"][:, 0]), "status f32", np.bincount(a["stats"][:, 5]))
    print("  fista F iters mean f64 %.0f f32 %.0f | X iters f64 %.0f f32 %.0f | bt f64 %s f32 %s" % (c["stats"][:, 1].mean(), a["stats"][:, 1].mean(), c["stats"][:, 2].mean(), a["stats"][:, 2].mean(), c["stats"][:, 3:5].mean(0), a["stats"][:, 3:5].mean(0)))
    print("  dyn_viol f64 med %.3e f32 med %.3e max %.3e" % (np.median(c["dyn_viol"]), np.median(a["dyn_viol"]), a["dyn_viol"].max()))

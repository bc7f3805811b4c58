#!/usr/bin/env python3
"""Where the fp32 kernel's tail comes from: per ADMM iteration, the F-step / X-step FISTA iteration counts of the fp32 solve against
the fp64 solve (bmpc_batch_t.trace) for the problems farthest from the fp64 result.  usage: tools/fp32_tail.py [config] [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import batch as bb, problems
cfg = sys.argv[1] if len(sys.argv) > 1 else "solo12_trot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
b = problems.make_batch(cfg, B)
r64 = bb.solve_host(b, num_iters=10, keep_hist=True)
r32 = bb.solve_host(b, num_iters=10, keep_hist=True, precision="f32")
e = np.maximum(np.linalg.norm(r32["X"] - r64["X"], axis=1) / np.linalg.norm(r64["X"], axis=1), np.linalg.norm(r32["F"] - r64["F"], axis=1) / np.linalg.norm(r64["F"], axis=1))
print("%s B=%d fp32 vs fp64: median %.2e p95 %.2e p99 %.2e max %.2e; above 1e-5: %d, above 1e-4: %d" % (cfg, B, np.median(e), np.quantile(e, .95), np.quantile(e, .99), e.max(), (e > 1e-5).sum(), (e > 1e-4).sum()))
def per_iter(tr):
    d = np.diff(np.concatenate([np.zeros((1, 4), dtype=np.int64), tr]), axis=0)
    return d
for i in np.argsort(-e)[:8]:
    d64, d32 = per_iter(r64["trace"][i]), per_iter(r32["trace"][i])
    print("problem %4d err %.2e  F-iters f64 %s" % (i, e[i], d64[:, 0].tolist()))
    print("                         F-iters f32 %s" % d32[:, 0].tolist())
    print("                         X-iters f64 %s" % d64[:, 1].tolist())
    print("                         X-iters f32 %s  retries f32 F %d X %d (f64 %d %d)" % (d32[:, 1].tolist(), r32["trace"][i, -1, 2], r32["trace"][i, -1, 3], r64["trace"][i, -1, 2], r64["trace"][i, -1, 3]))
    print("                         viol f64 %s" % " ".join("%.3e" % v for v in r64["hist"][i]))
    print("                         viol f32 %s" % " ".join("%.3e" % v for v in r32["hist"][i]))
same = np.all(r64["trace"] == r32["trace"], axis=(1, 2))
print("problems with the fp64 discrete path in fp32: %d of %d; their max err %.2e; the others' median err %.2e" % (same.sum(), B, e[same].max() if same.any() else 0, np.median(e[~same]) if (~same).any() else 0))

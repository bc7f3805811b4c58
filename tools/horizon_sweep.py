#!/usr/bin/env python3
"""The centroidal solve over horizons, as the reference's examples/analysis/solve_times_test.py sweeps them (gait horizons of 1 .. 10 s at
dt = 0.05 s): ms per batch solve of B problems on the GPU (10 ADMM iterations), the lanes / waves per problem the dispatch took, and the
CPU port (oracle, all host cores) on a sample.  usage: tools/horizon_sweep.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bunmpc_amd import _lib, batch as bb, problems
from oracle import oracle_c
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _lib.lib()
print("%5s %6s | %10s %12s %6s | %14s %8s" % ("H", "B", "GPU ms", "solves/s", "lanes", "CPU solves/s", "ratio"))
for H in (20, 40, 60, 63, 64, 80, 100, 127, 128, 160, 200, 255):
    pb = problems.make_batch("solo12_trot", B, H=H)
    db = bb.DeviceBatch(pb, num_iters=10)
    for _ in range(2):
        db.solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        db.solve()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    lanes = lib.bmpc_biconvex_last_lanes_per_problem()
    sub = pb.take(np.arange(0, B, max(1, B // 32))[:32])
    t0 = time.perf_counter()
    oracle_c.solve_batch(sub, num_iters=10)
    cpu = sub.B / (time.perf_counter() - t0)
    print("%5d %6d | %10.2f %12.3e %6d | %14.1f %8.0f" % (H, B, ms, B / ms * 1e3, lanes, cpu, B / ms * 1e3 / cpu), flush=True)
    del db

#!/usr/bin/env python3
"""the centroidal solve at num_iters = 100 (B = 4096 Solo12 trot): two per wave, three per wave, three per wave with work stealing"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bunmpc_amd import _lib, batch as bb, problems
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pb = problems.make_batch("solo12_trot", B)
cases = [("two per wave", 0, 0, 0, 0), ("two per wave, two waves per SIMD", 0, 0, 1, 0), ("three per wave", 1, 0, 0, 0), ("three per wave, two waves per SIMD", 1, 0, 1, 0),
         ("three per wave + work stealing", 1, 1, 0, 0)]
cases += [("stealing, two waves per SIMD, grid %d" % g, 1, 1, 1, g) for g in (1024, 1152, 1280, 1366, 1536, 2048)]
for name, three, steal, w2, grid in cases:
    lib.bmpc_set_three_per_wave(three); lib.bmpc_set_work_stealing(steal); lib.bmpc_set_two_waves_per_simd(w2); lib.bmpc_set_steal_grid(grid)
    db = bb.DeviceBatch(pb, num_iters=100)
    for _ in range(2):
        db.solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        db.solve()
    torch.cuda.synchronize()
    r = db.results()
    print("%-44s %.2f ms per launch (%s); ADMM iterations sum %d" % (name, (time.perf_counter() - t0) / 5 * 1e3, lib.bmpc_biconvex_last_kernel_name().decode(), r["stats"][:, 0].sum()))
lib.bmpc_set_three_per_wave(2); lib.bmpc_set_work_stealing(1); lib.bmpc_set_two_waves_per_simd(2); lib.bmpc_set_steal_grid(0)

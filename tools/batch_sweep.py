#!/usr/bin/env python3
"""Headline kernel over batch sizes: ms per launch, solves/s and the lanes per problem the dispatch took (bmpc_set_three_per_wave
mode 2), beside the two forced mappings.  usage: tools/batch_sweep.py [config]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bunmpc_amd import _lib, batch as bb, problems
cfg = sys.argv[1] if len(sys.argv) > 1 else "solo12_trot"
lib = _lib.lib()
old_lat = lib.bmpc_set_latency_mapping_max_batch(0)       # the one-knot-per-lane kernel at every size
print("%6s | %-28s | %-20s | %-20s" % ("B", "auto: ms  solves/s  lanes", "two per wave: ms", "three per wave: ms"))
for B in (1024, 2048, 3072, 4096, 5120, 6144, 8192, 9216, 12288):
    pb = problems.make_batch(cfg, B)
    row = []
    for mode in (2, 0, 1):
        lib.bmpc_set_three_per_wave(mode)
        db = bb.DeviceBatch(pb, num_iters=10)
        for _ in range(3):
            db.solve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            db.solve()
        torch.cuda.synchronize()
        row.append(((time.perf_counter() - t0) / 10 * 1e3, lib.bmpc_biconvex_last_lanes_per_problem()))
        del db
    print("%6d | %6.2f ms %9.3e  %2d        | %6.2f ms %9.3e | %6.2f ms %9.3e" % (B, row[0][0], B / row[0][0] * 1e3, row[0][1], row[1][0], B / row[1][0] * 1e3, row[2][0], B / row[2][0] * 1e3))
lib.bmpc_set_three_per_wave(2)
lib.bmpc_set_latency_mapping_max_batch(old_lat)

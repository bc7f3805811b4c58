#!/usr/bin/env python3
"""The fp64 batch kernel's two builds side by side (bmpc_set_two_waves_per_simd 0 / 1): ms per launch on the headline batch and on
the Go2 bound batch (64 lanes per problem).  usage: [BUNMPC_LIB=variant.so] tools/w2_timing.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bunmpc_amd import _lib, batch as bb, problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.lib()
lib.bmpc_set_latency_mapping_max_batch(0)
lib.bmpc_set_three_per_wave(0)
for cfg in ("solo12_trot", "go2_bound"):
    pb = problems.make_batch(cfg, B)
    row = []
    for mode in (0, 1):
        lib.bmpc_set_two_waves_per_simd(mode)
        db = bb.DeviceBatch(pb, num_iters=10)
        for _ in range(3):
            db.solve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            db.solve()
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 15 * 1e3)
        assert lib.bmpc_biconvex_last_waves_per_simd() == mode + 1
        del db
    print("%-12s B=%d H=%d: one wave per SIMD %.3f ms, two %.3f ms (%s)" % (cfg, B, pb.H, row[0], row[1], os.environ.get("BUNMPC_LIB", "default")))

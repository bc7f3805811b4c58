#!/usr/bin/env python3
"""fp32 kernel against the CPU oracle, problem by problem (GPU box).  BUNMPC_LIB=<side build> compares kernel versions."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import batch as bb, problems
from oracle import oracle_c
from tests.util import rel_l2
cfg, B = sys.argv[1], int(sys.argv[2])
b = problems.make_batch(cfg, B)
ref = oracle_c.solve_batch(b, num_iters=10)
for prec in ("f64", "f32"):
    got = bb.solve_host(b, num_iters=10, precision=prec)
    eX, eF = rel_l2(got["X"], ref["X"]), rel_l2(got["F"], ref["F"])
    bad = np.where(np.maximum(eX, eF) > 1e-4)[0]
    print(prec, "lib", os.environ.get("BUNMPC_LIB", "default"), "median", np.median(np.maximum(eX, eF)), "max", np.maximum(eX, eF).max(), "bad", bad)
    for i in bad[:6]:
        print("   ", i, "eX %.2e eF %.2e" % (eX[i], eF[i]), "gpu stats", got["stats"][i], "ref stats", ref["stats"][i], "dyn_viol", got["dyn_viol"][i])

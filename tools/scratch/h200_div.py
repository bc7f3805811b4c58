import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from bunmpc_amd import batch as bb, problems
from oracle import oracle_c
for H in (63, 100, 200):
    b = problems.make_batch("solo12_trot", 24, H=H)
    got = bb.solve_host(b, num_iters=10, keep_hist=True)
    ref = oracle_c.solve_batch(b, num_iters=10, trace=True)
    fast = oracle_c.solve_batch(b, num_iters=10, fast=True)
    print("H", H, "GPU status", got["stats"][:, 5].tolist())
    print("      strict     ", ref["stats"][:, 5].tolist())
    print("      matrix-free", fast["stats"][:, 5].tolist())
    print("      admm iters GPU", got["stats"][:, 0].tolist())
    print("      admm iters CPU", ref["stats"][:, 0].tolist())
    with np.errstate(all="ignore"):
        print("      hist GPU[0]", np.array2string(got["hist"][0], precision=3), "\n      hist CPU[0]", np.array2string(ref["hist"][0], precision=3))

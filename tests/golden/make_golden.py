"""Generates tests/golden/*.npz: seeded inputs (bunmpc_amd.problems) and the outputs of the
C restatement (oracle/biconvex_oracle.c) on them.

These are NOT reference outputs: the reference ships no vectors for this path and cannot be
built or imported here -- parity unpinned.  The fixtures freeze the oracle's
behaviour so that (a) a later edit of the oracle or of the problem generator cannot drift
silently and (b) the GPU tests have inputs/outputs that do not depend on building the oracle.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from bunmpc_amd import problems  # noqa: E402
from oracle import oracle_c  # noqa: E402

CASES = [  # name, config, B, num_iters
    ("solo12_trot_nominal_b1_it10", "solo12_trot_nominal", 1, 10),
    ("solo12_trot_b4_it10", "solo12_trot", 4, 10),
    ("solo12_mixed_b6_it1", "solo12_mixed", 6, 1),
    ("go2_bound_b2_it3", "go2_bound", 2, 3),
]


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, config, B, iters in CASES:
        b = problems.make_batch(config, B)
        r = oracle_c.solve_batch(b, num_iters=iters)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), config=config, num_iters=iters, m=b.m, rho=b.rho,
            mu=b.mu, cnt_plan=b.cnt_plan, dt=b.dt, x_init=b.x_init, X_nom=b.X_nom, X_ter=b.X_ter,
            W_X=b.W_X, W_X_ter=b.W_X_ter, W_F=b.W_F, bounds=b.bounds,
            X=r["X"], F=r["F"], P=r["P"], L_x=r["L_x"], L_f=r["L_f"], stats=r["stats"])
        print(name, "stats", r["stats"].tolist())


if __name__ == "__main__":
    main()

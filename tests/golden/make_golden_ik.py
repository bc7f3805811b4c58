"""Generates tests/golden/ik_*.npz: the inputs of whole-body IK-DDP solves exactly as bmpc_ik_solve_batch_device takes them
(x0, dt, task blocks WITH the CoM / momentum tracking references KinoDynMP::optimize fills from the centroidal solution,
regularisation weights and references) and the outputs of the compiled CPU twin (oracle/ik_ddp_oracle.c) on them: iteration
count, status, the per-iteration trace [cost, regularisation, accepted step length, stopping criterion], xs, us, cost.

NOT reference outputs: crocoddyl 1.9.0 / pinocchio 2.6.9 are absent here and the reference holds no vectors for this path
 -- parity unpinned.  The fixtures freeze the twin, the problem generator and the URDF-derived models together, so
that a drift shared by generator, twin and kernel cannot pass silently, and give the GPU tests a case that needs no oracle
build.  The Go2 fixture (synthetic robot, H = 60 / H_ik = 30) contains a problem that runs into SolverDDP's maxiter = 100.

Run from the repo root:  python tests/golden/make_golden_ik.py
"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from bunmpc_amd import problems, urdf_model  # noqa: E402
from oracle import ik_oracle_c as ic, oracle_c  # noqa: E402

ROBOTS = os.path.join(ROOT, "bunmpc_amd", "robots")


def wb_batch(robot, B):
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, robot + ".json")).read())
    if robot == "go2":
        wb = problems.make_wb_batch(model, B, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    else:
        wb = problems.make_wb_batch(model, B)
    return model, wb


CASES = [("ik_solo12_h20_b4", "solo12", 4), ("ik_go2_h60_b3", "go2", 3)]   # name, robot, B (Go2 problem 2 runs to maxiter)


def ik_inputs(model, wb, X):
    """the arrays of one bmpc_ik_batch_t, references filled as kd_fill_refs / kino_dyn.cpp:50-56 do"""
    T = wb.ik_T
    tasks = np.array(wb.ik_tasks, dtype=np.float64)
    Xk = np.asarray(X).reshape(wb.dyn.B, wb.dyn.H + 1, 9)[:, :T + 1]
    tasks[:, :, 21:24] = Xk[:, :, 0:3]
    tasks[:, :, 25:28] = wb.dyn.m * Xk[:, :, 3:6]
    tasks[:, :, 28:31] = Xk[:, :, 6:9]
    return dict(x0=np.array(wb.x), dt=np.array(wb.dyn.dt[:, :T]), tasks=tasks, state_w=np.array(wb.state_w), x_reg=np.array(wb.x_reg),
                ctrl_w=np.array(wb.ctrl_w))


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, robot, B in CASES:
        model, wb = wb_batch(robot, B)
        m = ic.Model(model)
        wb.dyn.x_init[:] = ic.centroidal_state(m, wb.x)
        X = oracle_c.solve_batch(wb.dyn, num_iters=10)["X"]
        inp = ik_inputs(model, wb, X)
        r = ic.solve_batch(m, inp["x0"], inp["dt"], inp["tasks"], inp["state_w"], inp["x_reg"], inp["ctrl_w"], trace=True)
        tr = np.nan_to_num(r["trace"], nan=0.0)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), robot=robot, T=wb.ik_T, H=wb.dyn.H, X=X, **inp,
                            iters=r["iters"], status=r["status"], cost=r["cost"], stop=r["stop"], xs=r["xs"], us=r["us"], trace=tr)
        print(name, "iters", r["iters"].tolist(), "status", r["status"].tolist())


if __name__ == "__main__":
    main()

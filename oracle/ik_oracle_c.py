"""ctypes access to oracle/liboracle_ik.so (oracle/ik_ddp_oracle.c: the compiled restatement of the
whole-body IK-DDP, twin of oracle/ik_ddp_np.py).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by bunmpc_amd.  PARITY UNPINNED (crocoddyl 1.9.0 / pinocchio 2.6.9 absent; see the C file)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ik.so")
TRACE = 4


def build(force=False):
    src = os.path.join(_HERE, "ik_ddp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_ik.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ikor_node.restype = C.c_double
        _lib.ikor_model_bytes.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Model:
    """ikor_model_t filled from a bunmpc_amd.urdf_model.RobotModel (frame ids = positions in model.frames)"""

    def __init__(self, model):
        self.model = model
        l = lib()
        self.buf = C.create_string_buffer(l.ikor_model_bytes())
        fr = list(model.frames.values())
        fbody = np.array([f[0] for f in fr], dtype=np.int32)
        fp = _f64([f[2] for f in fr])
        parent = np.ascontiguousarray(model.parent, dtype=np.int32)
        a = [_f64(x) for x in (model.R.reshape(model.nj, 9), model.p, model.axis, model.mass, model.com, model.inertia.reshape(-1, 9))]
        l.ikor_model_fill(self.buf, model.nj, _p(parent), *[_p(x) for x in a], len(fr), _p(fbody), _p(fp))
        self.nq, self.nv = model.nq, model.nv


def _strides(a, per_node_len, B, nn):
    """batch / node strides of a regularisation array shaped [len], [B or 1][len] or [B or 1][nn][len]"""
    a = _f64(a)
    if a.ndim == 1:
        return a, 0, 0
    if a.ndim == 2:
        return a, (0 if a.shape[0] == 1 else per_node_len), 0
    return a, (0 if a.shape[0] == 1 else nn * per_node_len), per_node_len


def solve_batch(m, x0, dt, tasks, state_w, x_reg, ctrl_w, maxiter=100, nthreads=0, trace=False):
    """B independent InverseKinematics::optimize calls on the arrays the GPU batch takes.
    x0 [B][nx], dt [B][T], tasks [B][T+1][33]; state_w [.][36], x_reg [B][37], ctrl_w [.][18]
    (or per-node variants [.][T+1][len])."""
    x0, dt, tasks = _f64(x0), _f64(dt), _f64(tasks)
    B, T = dt.shape
    nn = T + 1
    nx, nv = m.nq + m.nv, m.nv
    sw, s_sw, sn_sw = _strides(state_w, 2 * nv, B, nn)
    xr, s_xr, sn_xr = _strides(x_reg, nx, B, nn)
    cw, s_cw, sn_cw = _strides(ctrl_w, nv, B, nn)
    xs = np.zeros((B, nn, nx)); us = np.zeros((B, T, nv))
    iters = np.zeros(B, dtype=np.int32); status = np.zeros(B, dtype=np.int32)
    cost = np.zeros(B); stop = np.zeros(B); reg = np.zeros(B)
    tr = np.full((B, maxiter, TRACE), np.nan) if trace else None
    lib().ikor_solve_batch(m.buf, B, T, maxiter, _p(x0), _p(dt), _p(tasks), _p(sw), C.c_long(s_sw), C.c_long(sn_sw),
                           _p(xr), C.c_long(s_xr), C.c_long(sn_xr), _p(cw), C.c_long(s_cw), C.c_long(sn_cw),
                           _p(xs), _p(us), _p(iters), _p(status), _p(cost), _p(stop), _p(reg), _p(tr), nthreads)
    out = dict(xs=xs, us=us, iters=iters.astype(np.int64), status=status.astype(np.int64), cost=cost, stop=stop, reg=reg)
    if trace:
        out["trace"] = tr
    return out


def centroidal_state(m, x):
    x = _f64(np.atleast_2d(x))
    out = np.zeros((x.shape[0], 9))
    lib().ikor_centroidal_state(m.buf, x.shape[0], _p(x), _p(out))
    return out


def kin_quantities(m, x):
    x = _f64(x)
    nv = m.nv
    com, hg = np.zeros(3), np.zeros(6)
    Ag, Jc, dh = np.zeros((6, nv)), np.zeros((3, nv)), np.zeros((6, nv))
    lib().ikor_kin_quantities(m.buf, _p(x), _p(com), _p(hg), _p(Ag), _p(Jc), _p(dh))
    return dict(com=com, hg=hg, Ag=Ag, Jc=Jc, dh=dh)


def frame(m, x, fid):
    x = _f64(x)
    pos, J = np.zeros(3), np.zeros((3, m.nv))
    lib().ikor_frame(m.buf, _p(x), int(fid), _p(pos), _p(J))
    return pos, J


def state_ops(m, x0, x1, dx):
    x0, x1, dx = _f64(x0), _f64(x1), _f64(dx)
    ndx = 2 * m.nv
    diff, Jl, xint, A6, B6 = np.zeros(ndx), np.zeros((6, 6)), np.zeros(m.nq + m.nv), np.zeros((6, 6)), np.zeros((6, 6))
    lib().ikor_state_ops(m.buf, _p(x0), _p(x1), _p(dx), _p(diff), _p(Jl), _p(xint), _p(A6), _p(B6))
    return dict(diff=diff, Jl=Jl, xint=xint, A6=A6, B6=B6)


def node(m, T, t, dt, tasks, state_w, x_reg, ctrl_w, x, u):
    """cost and dense derivatives of node t (tasks [T+1][33])"""
    nv = m.nv
    ndx = 2 * nv
    dt, tasks, state_w, x_reg, ctrl_w, x = [_f64(a) for a in (dt, tasks, state_w, x_reg, ctrl_w, x)]
    u = _f64(u) if u is not None else np.zeros(nv)
    xn = np.zeros(m.nq + nv)
    Lx, Lxx, Lu, Luu = np.zeros(ndx), np.zeros((ndx, ndx)), np.zeros(nv), np.zeros((nv, nv))
    Fx, Fu = np.zeros((ndx, ndx)), np.zeros((ndx, nv))
    c = lib().ikor_node(m.buf, T, t, _p(dt), _p(tasks), _p(state_w), _p(x_reg), _p(ctrl_w), _p(x), _p(u), _p(xn), _p(Lx), _p(Lxx),
                        _p(Lu), _p(Luu), _p(Fx), _p(Fu))
    return dict(cost=c, xnext=xn, Lx=Lx, Lxx=Lxx, Lu=Lu, Luu=Luu, Fx=Fx, Fu=Fu)


def solve_wb_batch(m, wb, X, maxiter=100, nthreads=0, trace=False):
    """the IK half of KinoDynMP::optimize on a bunmpc_amd.problems.WholeBodyBatch: tracking references from the centroidal
    solution X [B][9(H+1)] (kino_dyn.cpp:50-56: rows 0..T-1 running, row T terminal; mom = [m v, L])"""
    T = wb.ik_T
    tasks = np.array(wb.ik_tasks, dtype=np.float64)
    Xk = np.asarray(X).reshape(wb.dyn.B, wb.dyn.H + 1, 9)[:, :T + 1]
    tasks[:, :, 21:24] = Xk[:, :, 0:3]
    tasks[:, :, 25:28] = wb.dyn.m * Xk[:, :, 3:6]
    tasks[:, :, 28:31] = Xk[:, :, 6:9]
    return solve_batch(m, wb.x, wb.dyn.dt[:, :T], tasks, wb.state_w, wb.x_reg, wb.ctrl_w, maxiter=maxiter, nthreads=nthreads, trace=trace)

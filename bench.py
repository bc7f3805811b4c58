#!/usr/bin/env python3
"""bench.py -- MPC solves/sec of the batched BiConvex MPC solve on MI355X.

One "step" = one pass of the hot path over one batch.  Default workload (`--workload biconvex`): B independent
BiconvexMP.optimize(x_init, 10) solves (Solo12 trot, H = 20, fp64, perturbed initial conditions; SURVEY.md 8d config 2 at
the batch size of north_star's target, 4096 per GPU) in ONE kernel launch, inputs already resident in HBM, cold start as
KinoDynMP::set_warm_starts does.  `--workload kinodyn` makes the full KinoDynMP.optimize (centroidal ADMM + whole-body
IK-DDP; BASELINE config 5's shape with `--kinodyn-config go2_h60`) the measured line instead.

Multi-GPU: one process per GPU, every leg shards its batch (rank r solves problems [r*B, (r+1)*B)), no data-path
collective (the solves are independent); RCCL only carries the timing / telemetry reductions.  Every leg is bracketed by a
barrier + synchronize on both sides and reports the MAX over ranks.  Under a launcher (the driver's torch.distributed.run line)
the ranks come from RANK / WORLD_SIZE; a plain `python bench.py --gpus N` starts that launcher itself, before this process
touches the GPU, and passes rank 0's line through.  At N = 1 RCCL is brought up with one rank as well (`rccl` in the line), so the
barriers and reductions of the N > 1 path run on every bench.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      HBM view of the dominant kernel (algorithmic bytes / measured kernel time; the path is fp64-VALU / latency
                bound, so `valu` carries the meaningful fraction) and
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, "port") timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N = 1 only).
The KinoDyn legs inside the line carry their own `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

# the CPU-baseline legs run OpenMP regions (oracle/*.so); their worker threads must go to sleep afterwards instead of spinning on
# the cores the host-driven DDP loop of the next GPU leg runs on (libgomp reads this when it is loaded)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # MI355X vector fp64 peak (spec)
METRIC_SHAPE = {"solo12_trot": "Solo12 trot H=20", "solo12_mixed": "Solo12 mixed gaits (trot/bound/pace) H=20", "go2_bound": "Go2 bound H=40",
                "solo12_trot_nominal": "Solo12 trot H=20 (nominal)"}


def flops_from_stats(stats, H, E=4):
    """SURVEY.md 8d flop model: F-step ~90 flops per knot.foot, X-step ~250 flops per knot,
    per FISTA iteration (retries counted as iterations)."""
    it_f = stats[:, 1].sum() + stats[:, 3].sum()
    it_x = stats[:, 2].sum() + stats[:, 4].sum()
    return float(it_f) * 90.0 * E * H + float(it_x) * 250.0 * (H + 1)


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def build_oracles():
    """the CPU checkers are rebuilt on the box that times them (-march=native must match its cores); the prebuilt files
    are used when no compiler is there"""
    from oracle import ik_oracle_c, oracle_c
    for mod in (oracle_c, ik_oracle_c):
        try:
            mod.build(force=True)
        except Exception:          # noqa: BLE001 -- the prebuilt library travelled with the tree
            mod.build()


def cpu_baseline(config, H_iters, sample, maxit):
    from bunmpc_amd import problems
    from oracle import oracle_c
    cores = host_cores()
    b = problems.make_batch(config, sample)
    oracle_c.solve_batch(b.slice(0, min(cores, sample)), num_iters=H_iters, maxit=maxit, nthreads=cores)  # warm-up
    t0 = time.perf_counter()
    oracle_c.solve_batch(b, num_iters=H_iters, maxit=maxit, nthreads=cores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    oracle_c.solve_batch(b.slice(0, 4), num_iters=H_iters, maxit=maxit, nthreads=1)
    lat = (time.perf_counter() - t1) / 4
    # the matrix-free CPU variant (oracle/biconvex_fast.c): same iteration, the structure of A_x / A_f exploited
    oracle_c.solve_batch(b.slice(0, min(cores, sample)), num_iters=H_iters, maxit=maxit, nthreads=cores, fast=True)
    t2 = time.perf_counter()
    oracle_c.solve_batch(b, num_iters=H_iters, maxit=maxit, nthreads=cores, fast=True)
    dt_fast = time.perf_counter() - t2
    return {"value": sample / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d problems of the workload, %d OpenMP threads, strict port (explicit sparse Hessian)" % (sample, cores),
            "single_core_ms_per_solve": lat * 1e3,
            "matrix_free_variant": {"value": sample / dt_fast, "unit": "solves/s", "cores": cores}}


def kinodyn_cpu_baseline(model, wb, admm_iters, maxit, sample):
    """KinoDynMP::optimize on the host: centroidal state of (q, v), the strict C restatement of the ADMM, the tracking
    references, then the compiled IK-DDP twin (oracle/ik_ddp_oracle.c: dense crocoddyl-shaped Riccati) -- one problem per
    OpenMP thread, on the first `sample` problems of the leg's batch."""
    from oracle import ik_oracle_c as ic, oracle_c
    cores = host_cores()
    sub = wb.take(np.arange(min(sample, wb.dyn.B)))
    m = ic.Model(model)
    n = sub.dyn.B

    def run(w, threads):
        t0 = time.perf_counter()
        w.dyn.x_init[:] = ic.centroidal_state(m, w.x)
        r = oracle_c.solve_batch(w.dyn, num_iters=admm_iters, maxit=maxit, nthreads=threads)
        t1 = time.perf_counter()
        k = ic.solve_wb_batch(m, w, r["X"], nthreads=threads)
        return t1 - t0, time.perf_counter() - t1, k
    run(sub.take(np.arange(min(cores, n))), cores)
    t_dyn, t_ik, k = run(sub, cores)
    l_dyn, l_ik, _ = run(sub.take(np.arange(min(2, n))), 1)
    return {"value": n / (t_dyn + t_ik), "unit": "KinoDynMP solves/s", "cores": cores, "kind": "port",
            "sample": "%d problems of the workload, %d OpenMP threads, strict ADMM port + compiled IK-DDP twin" % (n, cores),
            "dyn_seconds": t_dyn, "ik_seconds": t_ik, "single_core_ms_per_solve": (l_dyn + l_ik) / min(2, n) * 1e3,
            "ddp_iters_mean": float(k["iters"].mean()), "ddp_not_converged": int((k["status"] != 0).sum())}


def p50_latency(config, num_iters, reps=200, warm=20):
    """Batch-1 wall time of BiconvexMP.optimize through the drop-in class, incl. H2D of the
    inputs and D2H of X/F/P (p50 over >= 200 repeats after 20 warm-ups)."""
    from bunmpc_amd import problems
    from bunmpc_amd.biconvex_mpc_cpp import BiconvexMP
    b = problems.make_batch(config, 1)
    H, E = b.H, b.E
    mp = BiconvexMP(b.m, H, E)
    mp.set_rho(b.rho)
    X0, F0, P0 = b.warm_start()
    ts = []
    for r in range(reps + warm):
        for i in range(H):
            mp.set_contact_plan(b.cnt_plan[0, i], b.dt[0, i])
        mp.create_bound_constraints(b.bounds[0], 15.0, 15.0, 15.0)
        mp.create_cost_X(b.W_X[0], b.W_X_ter[0], b.X_ter[0], b.X_nom[0])
        mp.create_cost_F(b.W_F[0])
        mp.set_warm_start_vars(X0[0], F0[0], P0[0])
        mp.set_step_constants(2.25e6, 506.25)
        t0 = time.perf_counter()
        mp.optimize(b.x_init[0], num_iters)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[warm:]) * 1e3
    return {"p50_ms": float(np.median(ts)), "p90_ms": float(np.quantile(ts, 0.9)), "reps": reps, "warmups": warm}


class quiet_stdout:
    """fd-level silence: KinoDynMP prints from C++ ("Initialized Kino-Dyn planner", the solve times) as the reference does,
    and the only thing this program may write to stdout is its JSON line"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        os.close(self.null)


def kinodyn_latency(reps=(60, 30), warm=5):
    """p50 of the reference's own call, kd.optimize(q, v, N, 1) (abstract_cyclic_gen.py:663; N = 100 there, 10 in the
    benchmark configs), through the drop-in harness SoloMpcGaitGen on one Solo12: the wall time of KinoDynMP.optimize
    itself (plan and cost set-up excluded, as the reference's solve_times[2] counts it) against the 50 ms replanning
    budget (simulation.py:44)."""
    from bunmpc_amd import problems, urdf_model
    from bunmpc_amd.cyclic_gen import SoloMpcGaitGen
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "solo12.json")).read())
    q0 = problems.SOLO12_Q0.copy()
    with quiet_stdout():
        return _kinodyn_latency(model, q0, reps, warm)


def _kinodyn_latency(model, q0, reps, warm):
    from bunmpc_amd import problems
    from bunmpc_amd.cyclic_gen import SoloMpcGaitGen
    gg = SoloMpcGaitGen(model, model, np.concatenate([q0, np.zeros(18)]), 0.05, q0)
    import types
    g, ik = problems.TROT, problems.TROT_IK          # motions/cyclic/solo12_trot.py:12-41 as a BiconvexMotionParams-shaped object
    gg.update_gait_params(types.SimpleNamespace(
        gait_period=g.gait_period, stance_percent=list(g.stance_percent), gait_dt=g.gait_dt, phase_offset=list(g.phase_offset),
        step_ht=g.step_ht, nom_ht=g.nom_ht, gait_horizon=g.gait_horizon, W_X=g.W_X, W_X_ter=g.W_X_ter, W_F=g.W_F, rho=g.rho,
        ori_correction=list(g.ori_correction), swing_wt=list(ik["swing_wt"]), cent_wt=list(ik["cent_wt"]), reg_wt=list(ik["reg_wt"]),
        state_wt=ik["state_wt"], ctrl_wt=list(ik["ctrl_wt"])), 0.0)
    gg.kd.compute_solve_times()
    out = {"budget_ms": 50.0}
    for N, n in zip((10, 100), reps):
        ts, tot = [], []
        for r in range(n + warm):
            q, v = q0.copy(), np.zeros(18)
            t0 = time.perf_counter()
            gg.optimize(q, v, round(0.05 * (r % 10), 3), np.array([0.2, 0.0, 0.0]), 0.0, dyn_iters=N)
            tot.append(time.perf_counter() - t0)
            ts.append(gg.kd.return_solve_times()[2])
        ts, tot = np.array(ts[warm:]) * 1e3, np.array(tot[warm:]) * 1e3
        out["N=%d" % N] = {"kd_optimize_p50_ms": float(np.median(ts)), "kd_optimize_p90_ms": float(np.quantile(ts, 0.9)),
                           "harness_call_p50_ms": float(np.median(tot)), "reps": n, "warmups": warm}
    return out


def pmc_traffic(workload_key):
    """(bytes, source): HBM bytes per launch (per batch solve for the KinoDyn legs) from the BUILDER's rocprofv3 PMC run of exactly
    this workload, replayed from the committed profiles/pmc_traffic.json -- not measured by this process (PMC collection needs the
    profiler around the program).  (None, reason) when no such measurement is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(workload_key)
    except OSError:
        return None, "profiles/pmc_traffic.json missing"
    if not e or e.get("traffic_bytes") is None:
        return None, "no committed PMC run for this workload"
    return e["traffic_bytes"], "replayed from profiles/pmc_traffic.json (builder's rocprofv3 --pmc run, %s); not measured by this process" % e.get("raw_log", "?")


def pmc_valu_busy(workload_key, waves_per_simd):
    """The SIMDs' VALU occupancy of the headline kernel from the builder's SQ counter run (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave
    x waves per SIMD), replayed like the traffic; None when the committed run was of another build."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(workload_key) or {}
    except OSError:
        return None
    return e.get("simd_valu_busy_frac") if e.get("waves_per_simd") == waves_per_simd else None


def ik_algorithmic_bytes(T, E=4):
    """SURVEY.md 8d, the IK part of one KinoDynMP.optimize: in q, v (37) + per-knot foot targets / flags 4 E H_ik,
    out xs 37 (H_ik + 1) + us 18 H_ik doubles"""
    return 8 * (37 + 4 * E * T + 37 * (T + 1) + 18 * T)


class Dist:
    """the rank bookkeeping every leg shares: barrier + synchronize brackets, MAX / SUM reductions of scalars"""

    def __init__(self, torch, dist, dev, world, rank, rehearsal, group=False):
        self.torch, self.dist, self.dev, self.world, self.rank, self.rehearsal = torch, dist, dev, world, rank, rehearsal
        self.group = group          # a process group is up (RCCL; gloo in the rehearsal): also at world size 1

    def sync(self):
        if self.group:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def reduce(self, values, op):
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device="cpu" if self.rehearsal else self.dev)
        if self.group:
            self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return [float(v) for v in t]

    def timed(self, fn, steps, warmup):
        """W untimed + exactly K timed calls of fn between barrier + synchronize brackets; seconds, MAX over ranks"""
        for _ in range(warmup):
            fn()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.sync()
        return self.reduce([time.perf_counter() - t0], "MAX")[0]


_POOLS = {}

# prose that used to sit inside the legs' objects (the driver's record keeps the END of the line: numbers go there, see summary())
NOTES = {
    "cpu_baseline": "kind 'port': the CPU restatement of the reference algorithm under oracle/ (strict = the reference's explicit-sparse-"
                    "Hessian formulation; matrix_free_variant = the same iteration written matrix-free), one solve per OpenMP thread; "
                    "NOT the reference binary (it needs Eigen / pinocchio / crocoddyl, absent here)",
    "parity": "unpinned: the reference holds no vectors for this path and cannot be built here; GPU vs independent CPU restatements with "
              "prefix parity + per-problem CPU ensembles (tests/util.py, tests/test_parity_envelope_gpu.py, tests/test_ik_gpu.py)",
    "go2": "synthetic Go2 legs run with mu = 10, not the reference's fixed mu = 1, at which the reference algorithm NaNs for a 15 kg robot "
           "(tests/test_biconvex_gpu.py::test_go2_at_the_references_mu_1_diverges_as_the_oracle)",
    "horizon_200": "informational leg at the far end of the reference's solve-time sweep (10 s horizons); `diverged` there is the reference ALGORITHM's: from "
                   "~100 knots on its squared-norm cone projection lets a third of these perturbed trot problems overflow to NaN within ten ADMM "
                   "iterations, in the CPU restatements as on the GPU (tools/scratch/h200_div.py; DESIGN.md 4)",
    "latency_batch1": "BiconvexMP.optimize(x_init, N) on one problem through the drop-in class: H2D of the inputs, one launch, D2H of X / F / P",
    "traffic": "roofline.traffic and roofline.valu.simd_busy_frac_pmc are replayed from profiles/pmc_traffic.json (see roofline.traffic_source; "
               "the latter: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave x waves per SIMD, profiles/r04_pmc_sq.txt), never measured by this process",
}


def summary(out):
    """The numbers a reader of the line's tail needs, flat and last: per KinoDyn leg ms per batch solve, speed-up over the CPU port
    on all host cores and the dominant kernel; the batch-1 latencies."""
    s = {"headline_ms_per_step": out.get("ms_per_step"), "headline_solves_per_s": out.get("value"),
         "headline_waves_per_simd": out.get("roofline", {}).get("waves_per_simd"),
         "batch_6144_solves_per_s": out.get("batch_6144", {}).get("value") if isinstance(out.get("batch_6144"), dict) else None,
         "horizon_200_solves_per_s": out.get("horizon_200", {}).get("value") if isinstance(out.get("horizon_200"), dict) else None,
         "headline_speedup_vs_cpu_all_cores": out.get("speedup_vs_cpu_all_cores"),
         "headline_speedup_vs_matrix_free_cpu": out.get("speedup_vs_matrix_free_cpu"), "p50_latency_ms_batch1": out.get("p50_latency_ms_batch1")}
    for key, short in (("kinodyn_full_solve", "kinodyn_solo12"), ("kinodyn_go2_h60", "kinodyn_go2_h60"), ("kinodyn_n100", "kinodyn_n100")):
        leg = out.get(key)
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            s[short + "_error"] = leg["error"]
            continue
        dk = leg.get("roofline", {}).get("dominant_kernel", {})
        s[short + "_ms_per_step"] = leg.get("ms_per_step")
        s[short + "_solves_per_s"] = leg.get("value")
        s[short + "_speedup_vs_cpu_all_cores"] = leg.get("speedup_vs_cpu_all_cores")
        s[short + "_dominant_kernel"] = dk.get("kernel")
        s[short + "_dominant_kernel_ms"] = dk.get("ms_per_solve_batch")
    lk = out.get("latency_kinodyn_dropin")
    if isinstance(lk, dict):
        for N in (10, 100):
            s["kd_optimize_p50_ms_N%d" % N] = lk.get("N=%d" % N, {}).get("kd_optimize_p50_ms")
    return s


def kinodyn_leg(D, B, admm_iters, maxit, config="solo12_h20", steps=3, warmup=1, n_streams=3, cpu_sample=0):
    """The full KinoDynMP.optimize (centroidal ADMM + whole-body IK-DDP) over B perturbed whole-body states per GPU, device
    resident, rank r owning problems [r B, (r + 1) B).  solo12_h20: Solo12 trot, H = 20, H_ik = 10; go2_h60: BASELINE
    config 5's shape (synthetic Go2, trot, H = 60, H_ik = 30)."""
    import dataclasses
    from bunmpc_amd import _lib, batch as bb, problems, urdf_model
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    torch, dev = D.torch, D.dev
    robot = "go2" if config == "go2_h60" else "solo12"
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
    first = D.rank * B
    if config == "go2_h60":
        wb = problems.make_wb_batch(model, B, first=first, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0),
                                    wb=problems.GO2_WB)
    else:
        wb = problems.make_wb_batch(model, B, first=first)
    kb = KinoDynDeviceBatch(wb, model, device=dev, num_iters=admm_iters, maxit=maxit)
    dt = D.timed(kb.solve, steps, warmup) / steps
    dt_ik = D.timed(kb.solve_ik_only, steps, 0) / steps
    # where the time of one batch solve goes: events around every kernel of the DDP loop (a separate, untimed pass)
    lib = _lib.lib()
    lib.bmpc_ik_set_profile(1)
    kb.solve_ik_only()
    torch.cuda.synchronize(dev)
    lib.bmpc_ik_set_profile(0)
    prof = (5 * __import__("ctypes").c_double)()
    lib.bmpc_ik_last_profile(prof)
    kern = dict(zip(("ik_state_kernel", "ik_calcdiff_kernel", "ik_backward_kernel", "ik_forward_kernel", "other"), [float(v) for v in prof]))
    r = kb.results()
    tele = D.reduce([r["ik_iters"].sum(), (r["ik_status"] != 0).sum(), (r["stats"][:, 5] != 0).sum()], "SUM")
    it_max = D.reduce([r["ik_iters"].max()], "MAX")[0]
    # n_streams batches in flight on as many HIP streams, one host thread each (bunmpc_amd/pipeline.py): the tail
    # iterations of one batch -- a few stragglers, most of the chip idle -- overlap the bulk phases of the others.
    # Whole-job throughput of a generator that keeps several batches going.
    from bunmpc_amd.pipeline import IN_FLIGHT_SCHEDULE, StreamPool
    kbs = [KinoDynDeviceBatch(wb, model, device=dev, num_iters=admm_iters, maxit=maxit, schedule=IN_FLIGHT_SCHEDULE if n_streams > 1 else None)
           for _ in range(n_streams)]
    if n_streams not in _POOLS:    # one pool per run: HIP spreads streams over a few hardware queues, and new streams per
        _POOLS[n_streams] = StreamPool(dev, n_streams)       # leg can land on one queue and serialise
    pool = _POOLS[n_streams]
    dt2 = D.timed(lambda: pool.run([k.solve for k in kbs]), steps, 1) / steps
    same = True
    for k in kbs:
        r2 = k.results()
        same = same and bool(np.array_equal(r2["xs"], r["xs"]) and np.array_equal(r2["ik_iters"], r["ik_iters"]))
    del kbs
    W = D.world
    T, H = wb.ik_T, wb.dyn.H
    abytes = (bb.algorithmic_bytes_per_solve(H, 4) + ik_algorithmic_bytes(T)) * B
    dom = max(("ik_calcdiff_kernel", "ik_backward_kernel", "ik_forward_kernel"), key=lambda k: kern[k])
    wkey = "kinodyn %s H=%d H_ik=%d B=%d admm_iters=%d" % (config, H, T, B, admm_iters)
    traffic, traffic_source = pmc_traffic(wkey)
    out = {"value": W * B / dt, "unit": "KinoDynMP solves/s", "workload": "%s H=%d H_ik=%d B=%d/GPU" % (config, H, T, B),
           "n_gpus": W, "batch": B, "global_batch": B * W, "ms_per_step": dt * 1e3, "steps": steps,
           "multi_stream": {"streams": n_streams, "value": W * n_streams * B / dt2, "unit": "KinoDynMP solves/s",
                            "ms_per_round_of_batches": dt2 * 1e3, "results_equal_single_stream": same},
           "ik_only_ms_per_step": dt_ik * 1e3, "ik_kernel_ms_per_solve": kern,
           "ddp_iters_mean": tele[0] / (B * W), "ddp_iters_max": int(it_max), "ddp_not_converged": int(tele[1]),
           "admm_diverged": int(tele[2]), "admm_iters_mean": float(r["stats"][:, 0].mean()), "admm_iters_cap": admm_iters,
           "parity": "unpinned",
           # HBM view of the whole solve (every kernel of one KinoDynMP.optimize batch): SURVEY 8d's per-solve bytes x B over the
           # wall time of a solve; `dominant_kernel` the same bytes' IK share over that kernel's summed launches.  The path is
           # latency / issue bound (EXPERIMENTS.md 9), so the fraction is tiny by construction.
           "roofline": {"bound": "hbm", "achieved": abytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": abytes / dt / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                        "scope": "all kernels of one batch solve", "algorithmic_bytes_per_solve_batch": abytes,
                        "dominant_kernel": {"kernel": dom, "ms_per_solve_batch": kern[dom],
                                            "achieved": ik_algorithmic_bytes(T) * B / (kern[dom] * 1e-3) / 1e9 if kern[dom] > 0 else None,
                                            "algorithmic_bytes": ik_algorithmic_bytes(T) * B}}}
    # the inputs of the same batch built on the device from the raw states (bmpc_wb_plan_batch_device)
    if config == "solo12_h20":
        from bunmpc_amd import fk_np
        from bunmpc_amd.inverse_kinematics_cpp import as_device_model
        from bunmpc_amd.plan_batch import DeviceWbPlan
        k0 = fk_np.kinematics(model, problems.SOLO12_Q0[None])
        offs = np.round(fk_np.frame_positions(model, k0, problems.HIPS)[0] - k0["com"][0], 3)
        offs[:, 1] += np.array([0.04, -0.04, 0.04, -0.04])
        p = DeviceWbPlan(as_device_model(model), problems.TROT, offs[:, :2], problems.FEET, problems.TROT_IK, wb.x, wb.dyn.meta["t0"],
                         wb.dyn.meta["v_des_body"], wb.dyn.H, wb.ik_T, device=dev)
        out["device_built_inputs_ms"] = D.timed(p.build, 10, 1) / 10 * 1e3
    if cpu_sample and D.world == 1:
        try:
            out["cpu_baseline"] = kinodyn_cpu_baseline(model, wb, admm_iters, maxit, cpu_sample)
            out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:       # noqa: BLE001 -- the GPU measurement above must not be lost with it
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def datagen_leg(dev, B, admm_iters):
    """Informational: one device-resident pass of the data path around the solve -- perturb B nominal
    Solo12 states, build their plans, KinoDynMP.optimize, 1 kHz plans, inverse-dynamics labels for the first 50 ms --
    with the stages timed apart by events (bunmpc_amd/datagen.py)."""
    import torch
    from bunmpc_amd import problems, urdf_model
    from bunmpc_amd.datagen import PlanLabelGenerator
    from bunmpc_amd.robot_id_controller import id_batch_device
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "solo12.json")).read())
    wb = problems.make_wb_batch(model, B)
    gen = PlanLabelGenerator(model, dyn_iters=admm_iters, device=dev)
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    # nominal states: the default stance at the batch's phases / desired velocities
    q = up(np.tile(problems.SOLO12_Q0, (B, 1)))
    v = torch.zeros((B, 18), dtype=torch.float64, device=dev)
    t0, vdes = up(wb.dyn.meta["t0"]), up(wb.dyn.meta["v_des_body"])
    g = torch.Generator(device=dev).manual_seed(7)
    out = gen.step(q, v, t0, vdes, generator=g)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    out = gen.step(q, v, t0, vdes, generator=g)
    torch.cuda.synchronize(dev)
    total = time.perf_counter() - t
    sol = out["solution"]
    R = out["states"].shape[1]

    def timed(fn, reps=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps
    z = torch.randn((B, gen.sampler.K, 36), dtype=torch.float64, device=dev, generator=g)
    contact = sol["plan"].cnt_plan[:, 0, :, 0]
    perturb_ms = timed(lambda: gen.sampler.apply(q, v, contact, z))
    xs = sol["xs_int"][:, :R].reshape(B * R, 37)
    us = sol["us_int"][:, :R].reshape(B * R, 18)
    f = sol["f_int"][:, :R].reshape(B * R, 12)
    id_ms = timed(lambda: id_batch_device(gen.mpc.dm, gen.foot_frames, gen.kp, gen.kd, xs[:, :19], xs[:, 19:], us, f, want=("action", "state")))
    id_bytes = B * R * ((37 + 18 + 12 + 37) + (12 + 43)) * 8
    return {"workload": "solo12 trot: perturb -> plan -> KinoDynMP.optimize -> 1 kHz plan -> ID labels", "batch": B,
            "label_rows": B * R, "ms_per_pass": total * 1e3, "label_rows_per_s": B * R / total,
            "perturb_kernel_ms": perturb_ms, "rejected_states": int(out["rejected"].numel()),
            "id_kernel_ms": id_ms, "id_rows_per_s": B * R / id_ms * 1e3, "id_algorithmic_GBps": id_bytes / id_ms / 1e6,
            "id_frac_of_hbm_peak": id_bytes / id_ms / 1e6 / 8000.0}


def biconvex_leg(D, args):
    """the headline: one launch of biconvex_admm_kernel over B problems per GPU per step"""
    import torch
    from bunmpc_amd import batch as bb
    from bunmpc_amd import problems
    B, dev = args.batch, D.dev
    pb = problems.make_batch(args.config, B, first=D.rank * B)
    db = bb.DeviceBatch(pb, device=dev, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)
    for _ in range(args.warmup):
        db.solve()
    D.sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        db.solve()
        ev[s][1].record()
    D.sync()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    res = db.results()
    elapsed, kern_ms = D.reduce([elapsed, kern_ms], "MAX")
    counts = D.reduce([(res["stats"][:, 5] != 0).sum(), res["stats"][:, 1:3].sum(), flops_from_stats(res["stats"], pb.H)], "SUM")
    W = D.world
    total = B * W * args.steps
    per_w = pb.W_X.shape[0] != 1
    abytes = bb.algorithmic_bytes_per_solve(pb.H, pb.E, per_w) * B
    achieved = abytes / (kern_ms * 1e-3) / 1e9
    flops = counts[2] / W  # per launch on one GPU
    wkey = "%s H=%d B=%d admm_iters=%d fista_maxit=%d %s" % (args.config, pb.H, B, args.admm_iters, args.maxit, args.precision)
    traffic, traffic_source = pmc_traffic(wkey)
    # the kernel the dispatch took for this batch (small batches of short horizons go to the one-problem-per-wave kernel)
    last_kernel = bb._lib.lib().bmpc_biconvex_last_kernel_name().decode()
    last_kernel = {"biconvex_admm_kernel": "biconvex_admm_kernel<double>"}.get(last_kernel, last_kernel)
    prec = {"f64": "fp64", "f32": "fp32 iterates, fp64 decisions"}[args.precision]
    lpp = int(bb._lib.lib().bmpc_biconvex_last_lanes_per_problem())
    wpe = int(bb._lib.lib().bmpc_biconvex_last_waves_per_simd())
    out = {
        "metric": "MPC solves/sec (batch, whole node), %s, %d ADMM iters, %s" % (METRIC_SHAPE.get(args.config, args.config), args.admm_iters, prec),
        "value": total / elapsed, "unit": "solves/s", "n_gpus": W, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": "%s H=%d B=%d/GPU admm_iters=%d fista_maxit=%d cold-start"
                               % (args.config, pb.H, B, args.admm_iters, args.maxit),
                   "global_batch": B * W, "parallelism": "batch-shard x%d" % W},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": last_kernel, "lanes_per_problem": lpp, "waves_per_simd": wpe, "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": abytes,
                     "valu": {"model_flops_per_launch": flops,
                              "achieved_tflops": flops / (kern_ms * 1e-3) / 1e12,
                              "peak_tflops": FP64_VALU_PEAK_TF,
                              "frac": flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                              "simd_busy_frac_pmc": pmc_valu_busy(wkey, wpe)}},
        "diverged": int(counts[0]), "fista_iters_per_solve": counts[1] / (B * W),
    }
    return out, pb


def other_batch_size_leg(D, args, B):
    """Informational: the same kernel at a batch size where three problems per wave fill the chip exactly (bmpc_set_three_per_wave,
    bmpc_set_two_waves_per_simd: B = 6144 is 2048 waves of three, two per SIMD)"""
    import torch
    from bunmpc_amd import batch as bb
    from bunmpc_amd import problems
    pb = problems.make_batch(args.config, B, first=D.rank * B)
    db = bb.DeviceBatch(pb, device=D.dev, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)
    dt = D.timed(db.solve, 10, 2) / 10
    r = db.results()
    return {"batch": B, "value": D.world * B / dt, "unit": "solves/s", "ms_per_step": dt * 1e3, "diverged": int((r["stats"][:, 5] != 0).sum()),
            "lanes_per_problem": int(bb._lib.lib().bmpc_biconvex_last_lanes_per_problem()),
            "waves_per_simd": int(bb._lib.lib().bmpc_biconvex_last_waves_per_simd())}


def long_horizon_leg(D, args, H=200, B=1024):
    """Informational: a horizon of the reference's solve-time sweep (examples/analysis/solve_times_test.py goes to 10 s = 200 knots): one
    problem per workgroup of four waves"""
    from bunmpc_amd import batch as bb
    from bunmpc_amd import problems
    pb = problems.make_batch(args.config, B, H=H, first=D.rank * B)
    db = bb.DeviceBatch(pb, device=D.dev, num_iters=args.admm_iters, maxit=args.maxit)
    dt = D.timed(db.solve, 5, 2) / 5
    r = db.results()
    return {"H": H, "batch": B, "value": D.world * B / dt, "unit": "solves/s", "ms_per_step": dt * 1e3, "diverged": int((r["stats"][:, 5] != 0).sum()),
            "kernel": bb._lib.lib().bmpc_biconvex_last_kernel_name().decode(), "lanes_per_problem": int(bb._lib.lib().bmpc_biconvex_last_lanes_per_problem()),
            "waves_per_simd": int(bb._lib.lib().bmpc_biconvex_last_waves_per_simd())}


def fp32_parity_note(pb, args):
    """config 3's residual check in the line: the fp32 kernel against the CPU oracle on a sample of the batch"""
    from bunmpc_amd import batch as bb
    from oracle import oracle_c
    sub = pb.take(np.arange(0, pb.B, max(1, pb.B // 64))[:64])
    ref = oracle_c.solve_batch(sub, num_iters=args.admm_iters, maxit=args.maxit)
    got = bb.solve_host(sub, num_iters=args.admm_iters, maxit=args.maxit, precision="f32")
    e = np.maximum(np.linalg.norm(got["X"] - ref["X"], axis=1) / np.linalg.norm(ref["X"], axis=1),
                   np.linalg.norm(got["F"] - ref["F"], axis=1) / np.linalg.norm(ref["F"], axis=1))
    return {"sample": int(sub.B), "rel_l2_vs_cpu_oracle_median": float(np.median(e)), "rel_l2_vs_cpu_oracle_max": float(e.max()),
            "same_admm_count": bool(np.array_equal(got["stats"][:, 0], ref["stats"][:, 0]))}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes the way the driver does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`), BEFORE this process has touched the GPU, pass
    rank 0's JSON line through and exit with the launcher's code.  The library is built (if stale) here, once, so that the
    ranks only load it."""
    import subprocess
    from bunmpc_amd import build as hip_build
    if os.path.exists(hip_build.HIPCC):
        hip_build.build()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BUNMPC_BENCH_CHILD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def init_rccl_at_one(torch, dist, dev):
    """World size 1 and no launcher: bring RCCL up anyway (one rank, TCP store on 127.0.0.1), so that the barriers and the
    MAX / SUM reductions of every leg run through the same collectives as at N > 1 -- and a broken RCCL shows at N = 1.
    Returns the record that goes into the line; a failure is recorded, never fatal (the N = 1 measurement does not need it)."""
    if os.environ.get("BUNMPC_BENCH_NO_RCCL_AT_ONE") == "1":
        return {"initialised": False, "note": "switched off (BUNMPC_BENCH_NO_RCCL_AT_ONE=1)"}
    try:
        import datetime
        t0 = time.perf_counter()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev,
                                timeout=datetime.timedelta(seconds=120))
        t = torch.tensor([3.0, 5.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.barrier()
        torch.cuda.synchronize(dev)
        ok = [float(v) for v in t] == [3.0, 5.0]
        return {"initialised": bool(ok), "backend": "nccl (RCCL)", "world": 1, "init_seconds": time.perf_counter() - t0}
    except Exception as e:       # noqa: BLE001 -- recorded in the line
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:        # noqa: BLE001
            pass
        return {"initialised": False, "error": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="biconvex", choices=["biconvex", "kinodyn"],
                    help="what the JSON line's metric / value measure: the centroidal batch solve (headline) or the full KinoDynMP.optimize")
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU")
    ap.add_argument("--config", default="solo12_trot", help="solo12_trot (headline) | solo12_mixed (BASELINE config 4) | go2_bound (config 3)")
    ap.add_argument("--admm-iters", type=int, default=10)
    ap.add_argument("--maxit", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-kinodyn", action="store_true")
    ap.add_argument("--kinodyn-main-only", action="store_true", help="skip the Go2 H=60 and data-path legs (counter collection)")
    ap.add_argument("--kinodyn-streams", type=int, default=3, help="batches in flight in the multi-stream KinoDyn measurement")
    ap.add_argument("--kinodyn-batch", type=int, default=0, help="problems per GPU of the KinoDyn leg (default 4096; 1024 for go2_h60)")
    ap.add_argument("--kinodyn-steps", type=int, default=3)
    ap.add_argument("--kinodyn-config", default="solo12_h20", choices=["solo12_h20", "go2_h60"],
                    help="solo12_h20: Solo12 trot H=20 / H_ik=10; go2_h60: BASELINE config 5 (synthetic Go2, H=60 / H_ik=30)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"],
                    help="f32: BASELINE config 3's mixed-precision kernel (fp32 iterates, fp64 decisions)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if os.environ.get("BUNMPC_BENCH_CHILD") == "1":
            raise SystemExit("bench.py: launched as a rank but WORLD_SIZE is missing")
        raise SystemExit(self_launch(args.gpus))

    # stdout carries the JSON line and nothing else: whatever the legs (or the C++ side of the drop-in classes) print goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the solve)")
    # rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0, gloo instead of RCCL (never used by the driver)
    rehearsal = os.environ.get("BUNMPC_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl = None
    if world > 1 or "WORLD_SIZE" in os.environ:      # under a launcher (torchrun sets WORLD_SIZE also for one rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        rccl = {"initialised": True, "backend": "gloo (one-device rehearsal)" if rehearsal else "nccl (RCCL)", "world": world}
    else:
        rccl = init_rccl_at_one(torch, dist, dev)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    D = Dist(torch, dist, dev, world, rank, rehearsal, group=dist.is_initialized())
    cpu_ok = world == 1 and not args.no_cpu
    if cpu_ok:
        build_oracles()
    kd_batch = lambda cfg: args.kinodyn_batch or (1024 if cfg == "go2_h60" else 4096)      # noqa: E731
    kd_sample = lambda cfg: 0 if not cpu_ok else (256 if cfg == "go2_h60" else 2048)      # noqa: E731

    def guarded(fn):
        """a failing secondary leg is reported in the line (never hidden, never allowed to lose the measurement already
        made); every rank takes part in every leg, so a leg's collectives stay matched"""
        try:
            return fn()
        except Exception as e:       # noqa: BLE001 -- recorded in the output
            return {"error": "%s: %s" % (type(e).__name__, e)}

    if args.workload == "kinodyn":
        cfg = args.kinodyn_config
        leg = kinodyn_leg(D, kd_batch(cfg), args.admm_iters, args.maxit, cfg, steps=args.steps, warmup=args.warmup,
                          n_streams=args.kinodyn_streams, cpu_sample=kd_sample(cfg))
        shape = "Go2 trot H=60 with full IK-DDP inner loop (H_ik=30)" if cfg == "go2_h60" else "Solo12 trot H=20 with full IK-DDP (H_ik=10)"
        out = {"metric": "KinoDynMP solves/sec (batch, whole node), %s, %d ADMM iters, fp64" % (shape, args.admm_iters),
               "value": leg["value"], "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": "kinodyn %s admm_iters=%d fista_maxit=%d ddp_maxiter=100 cold-start" % (leg["workload"], args.admm_iters, args.maxit),
                          "global_batch": leg["global_batch"], "parallelism": "batch-shard x%d" % world},
               "roofline": leg.pop("roofline")}
        if "cpu_baseline" in leg:
            out["cpu_baseline"] = leg.pop("cpu_baseline")
            out["speedup_vs_cpu_all_cores"] = leg.pop("speedup_vs_cpu_all_cores", None)
        out["details"] = leg
    else:
        out, pb = biconvex_leg(D, args)
        if cpu_ok:
            try:
                out["cpu_baseline"] = cpu_baseline(args.config, args.admm_iters, args.cpu_sample, args.maxit)
                out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
                out["speedup_vs_matrix_free_cpu"] = out["value"] / out["cpu_baseline"]["matrix_free_variant"]["value"]
            except Exception as e:       # noqa: BLE001 -- the GPU measurement above must not be lost with it
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if args.precision == "f32":
                out["fp32_residual_check"] = guarded(lambda: fp32_parity_note(pb, args))

        def host_leg():
            # the same batch through the host-buffer entry point (pageable numpy arrays in, H2D, one launch, D2H):
            # the PCIe-inclusive rate -- reported beside `value`, never as it
            from bunmpc_amd import batch as bb
            bb.solve_host(pb, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)
            th = time.perf_counter()
            for _ in range(3):
                bb.solve_host(pb, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)
            return 3 * pb.B / (time.perf_counter() - th)
        if args.config == "solo12_trot" and args.precision == "f64":
            out["batch_6144"] = guarded(lambda: other_batch_size_leg(D, args, 6144))
            out["horizon_200"] = guarded(lambda: long_horizon_leg(D, args))
        if world == 1 and not args.no_latency:
            lat = guarded(lambda: p50_latency(args.config, args.admm_iters))
            out["p50_latency_ms_batch1"] = lat.get("p50_ms", lat)
            out["latency_batch1"] = lat
            out["latency_kinodyn_dropin"] = guarded(kinodyn_latency)
            out["host_buffers_solves_per_s"] = guarded(host_leg)
        if not args.no_kinodyn:
            cfg = args.kinodyn_config
            out["kinodyn_full_solve"] = guarded(lambda: kinodyn_leg(D, kd_batch(cfg), args.admm_iters, args.maxit, cfg, steps=args.kinodyn_steps,
                                                                     n_streams=args.kinodyn_streams, cpu_sample=kd_sample(cfg)))
            if cfg == "solo12_h20" and not args.kinodyn_main_only:   # BASELINE config 5's shape as well (1024 problems = its per-GPU share)
                out["kinodyn_go2_h60"] = guarded(lambda: kinodyn_leg(D, kd_batch("go2_h60"), args.admm_iters, args.maxit, "go2_h60",
                                                                      steps=args.kinodyn_steps, n_streams=args.kinodyn_streams,
                                                                      cpu_sample=kd_sample("go2_h60")))
                if world == 1:
                    out["datagen_pass"] = guarded(lambda: datagen_leg(dev, kd_batch(cfg), args.admm_iters))
                # the reference's OWN call: kd.optimize(q, v, 100, 1) (abstract_cyclic_gen.py:663) at the headline's batch size
                out["kinodyn_n100"] = guarded(lambda: kinodyn_leg(D, kd_batch(cfg), 100, args.maxit, cfg, steps=args.kinodyn_steps, n_streams=1,
                                                                  cpu_sample=512 if cpu_ok else 0))
    out["rccl"] = rccl
    out["notes"] = NOTES
    out["summary"] = summary(out)
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- MPC solves/sec of the batched BiConvex MPC solve on MI355X.

One "step" = one pass of the hot path over one batch: B independent
BiconvexMP.optimize(x_init, 10) solves (Solo12 trot, H = 20, fp64, perturbed initial
conditions; SURVEY.md 8d config 2 at the batch size of north_star's target, 4096 per GPU)
in ONE kernel launch, inputs already resident in HBM, cold start as
KinoDynMP::set_warm_starts does.  Multi-GPU: one process per GPU, the batch is sharded
(rank r solves problems [r*B, (r+1)*B)), no data-path collective (the solves are
independent); RCCL only carries the timing/telemetry reductions.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      HBM view of the ADMM kernel (algorithmic bytes per launch / measured kernel
                time; the path is fp64-VALU/latency bound, so `valu` carries the meaningful
                fraction) and
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, "port") timed on
                this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # MI355X vector fp64 peak (spec)


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


def cpu_baseline(config, H_iters, sample, maxit):
    from bunmpc_amd import problems
    from oracle import oracle_c
    oracle_c.build()
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
            "sample": "%d problems of the same workload, one solve per OpenMP thread over %d threads, "
                      "strict restatement of the reference formulation (explicit sparse Hessian); "
                      "not the reference binary (needs Eigen, absent)" % (sample, cores),
            "single_core_ms_per_solve": lat * 1e3,
            "matrix_free_variant": {"value": sample / dt_fast, "unit": "solves/s", "cores": cores,
                                    "note": "same algorithm written matrix-free for the CPU (oracle/biconvex_fast.c); "
                                            "the reference itself is the explicit-Hessian formulation above"}}


def p50_latency(config, num_iters, reps=60):
    """Batch-1 wall time of BiconvexMP.optimize through the drop-in class, incl. H2D of the
    inputs and D2H of X/F/P."""
    from bunmpc_amd import problems
    from bunmpc_amd.biconvex_mpc_cpp import BiconvexMP
    b = problems.make_batch(config, 1)
    H, E = b.H, b.E
    mp = BiconvexMP(b.m, H, E)
    mp.set_rho(b.rho)
    X0, F0, P0 = b.warm_start()
    ts = []
    for r in range(reps + 5):
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
    return float(np.median(ts[5:]) * 1e3)


def pmc_traffic(workload_key):
    """HBM bytes per launch measured with rocprofv3 PMC counters for exactly this workload
    (profiles/pmc_traffic.json), or None when no such measurement is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(workload_key, {}).get("traffic_bytes")
    except OSError:
        return None


_POOLS = {}


def kinodyn_leg(dev, B, admm_iters, maxit, config="solo12_h20", steps=3, n_streams=3):
    """Informational: the full KinoDynMP.optimize (centroidal ADMM + whole-body IK-DDP) over B perturbed
    whole-body states, device resident.  solo12_h20: Solo12 trot, H = 20, H_ik = 10;
    go2_h60: BASELINE config 5's shape (synthetic Go2, trot, H = 60, H_ik = 30)."""
    import dataclasses
    import torch
    from bunmpc_amd import problems, urdf_model
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    robot = "go2" if config == "go2_h60" else "solo12"
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", robot + ".json")).read())
    if config == "go2_h60":
        wb = problems.make_wb_batch(model, B, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0),
                                    wb=problems.GO2_WB)
    else:
        wb = problems.make_wb_batch(model, B)
    kb = KinoDynDeviceBatch(wb, model, device=dev, num_iters=admm_iters, maxit=maxit)
    kb.solve()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        kb.solve()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    t1 = time.perf_counter()
    for _ in range(steps):
        kb.solve_ik_only()
    torch.cuda.synchronize(dev)
    dt_ik = (time.perf_counter() - t1) / steps
    r = kb.results()
    # n_streams batches in flight on as many HIP streams, one host thread each (bunmpc_amd/pipeline.py): the tail
    # iterations of one batch -- a few stragglers, most of the chip idle -- overlap the bulk phases of the others.
    # Whole-job throughput of a generator that keeps several batches going.
    from bunmpc_amd.pipeline import StreamPool
    kbs = [kb] + [KinoDynDeviceBatch(wb, model, device=dev, num_iters=admm_iters, maxit=maxit) for _ in range(n_streams - 1)]
    for k in kbs[1:]:
        k.solve()
    torch.cuda.synchronize(dev)
    if n_streams not in _POOLS:    # one pool per run: HIP spreads streams over a few hardware queues, and new streams per
        _POOLS[n_streams] = StreamPool(dev, n_streams)       # leg can land on one queue and serialise
    t3 = time.perf_counter()
    _POOLS[n_streams].run([k.solve for _ in range(steps) for k in kbs])
    torch.cuda.synchronize(dev)
    dt2 = (time.perf_counter() - t3) / steps
    same = True
    for k in kbs[1:]:
        r2 = k.results()
        same = same and bool(np.array_equal(r2["xs"], r["xs"]) and np.array_equal(r2["ik_iters"], r["ik_iters"]))
    del kbs
    # the inputs of the same batch built on the device from the raw states (bmpc_wb_plan_batch_device)
    plan_ms = None
    if config == "solo12_h20":
        from bunmpc_amd import fk_np
        from bunmpc_amd.inverse_kinematics_cpp import as_device_model
        from bunmpc_amd.plan_batch import DeviceWbPlan
        k0 = fk_np.kinematics(model, problems.SOLO12_Q0[None])
        offs = np.round(fk_np.frame_positions(model, k0, problems.HIPS)[0] - k0["com"][0], 3)
        offs[:, 1] += np.array([0.04, -0.04, 0.04, -0.04])
        p = DeviceWbPlan(as_device_model(model), problems.TROT, offs[:, :2], problems.FEET, problems.TROT_IK, wb.x, wb.dyn.meta["t0"],
                         wb.dyn.meta["v_des_body"], wb.dyn.H, wb.ik_T, device=dev)
        p.build()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(10):
            p.build()
        torch.cuda.synchronize(dev)
        plan_ms = (time.perf_counter() - t2) / 10 * 1e3
    return {"value": B / dt, "unit": "KinoDynMP solves/s", "workload": "%s H=%d H_ik=%d" % (config, wb.dyn.H, wb.ik_T),
            "device_built_inputs_ms": plan_ms,
            "batch": B, "ms_per_step": dt * 1e3,
            "multi_stream": {"streams": n_streams, "value": n_streams * B / dt2, "unit": "KinoDynMP solves/s",
                             "ms_per_round_of_batches": dt2 * 1e3, "results_equal_single_stream": same},
            "ik_only_ms_per_step": dt_ik * 1e3, "ddp_iters_mean": float(r["ik_iters"].mean()),
            "ddp_iters_max": int(r["ik_iters"].max()), "ddp_not_converged": int((r["ik_status"] != 0).sum()),
            "admm_diverged": int((r["stats"][:, 5] != 0).sum())}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU")
    ap.add_argument("--config", default="solo12_trot")
    ap.add_argument("--admm-iters", type=int, default=10)
    ap.add_argument("--maxit", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-kinodyn", action="store_true")
    ap.add_argument("--kinodyn-main-only", action="store_true", help="skip the Go2 H=60 and data-path legs (counter collection)")
    ap.add_argument("--kinodyn-streams", type=int, default=3, help="batches in flight in the multi-stream KinoDyn measurement")
    ap.add_argument("--kinodyn-batch", type=int, default=4096)
    ap.add_argument("--kinodyn-config", default="solo12_h20", choices=["solo12_h20", "go2_h60"],
                    help="solo12_h20: Solo12 trot H=20 / H_ik=10; go2_h60: BASELINE config 5 (synthetic Go2, H=60 / H_ik=30)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"],
                    help="f32: BASELINE config 3's mixed-precision kernel (fp32 iterates, fp64 decisions)")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    from bunmpc_amd import batch as bb
    from bunmpc_amd import problems
    B = args.batch
    pb = problems.make_batch(args.config, B, first=rank * B)
    db = bb.DeviceBatch(pb, device=dev, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        db.solve()
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        db.solve()
        ev[s][1].record()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    res = db.results()
    tele = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    counts = torch.tensor([float((res["stats"][:, 5] != 0).sum()), float(res["stats"][:, 1:3].sum()),
                           flops_from_stats(res["stats"], pb.H)], dtype=torch.float64, device=dev)
    if world > 1:
        if rehearsal:
            tele, counts = tele.cpu(), counts.cpu()
        dist.all_reduce(tele, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed, kern_ms = float(tele[0]), float(tele[1])
    total = B * world * args.steps

    if rank == 0:
        per_w = pb.W_X.shape[0] != 1
        abytes = bb.algorithmic_bytes_per_solve(pb.H, pb.E, per_w) * B
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        flops = float(counts[2]) / world  # per launch on one GPU
        wkey = "%s H=%d B=%d admm_iters=%d fista_maxit=%d %s" % (args.config, pb.H, B, args.admm_iters, args.maxit, args.precision)
        traffic = pmc_traffic(wkey)
        out = {
            "metric": "MPC solves/sec (batch, whole node), Solo12 trot H=20, 10 ADMM iters, fp64",
            "value": total / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": "%s H=%d B=%d/GPU admm_iters=%d fista_maxit=%d cold-start"
                                   % (args.config, pb.H, B, args.admm_iters, args.maxit),
                       "global_batch": B * world, "parallelism": "batch-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "biconvex_admm_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "valu": {"model_flops_per_launch": flops,
                                  "achieved_tflops": flops / (kern_ms * 1e-3) / 1e12,
                                  "peak_tflops": FP64_VALU_PEAK_TF,
                                  "frac": flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF}},
            "diverged": int(counts[0]), "fista_iters_per_solve": float(counts[1]) / (B * world),
        }
        if world == 1 and not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args.config, args.admm_iters, args.cpu_sample, args.maxit)
                out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
                out["speedup_vs_matrix_free_cpu"] = out["value"] / out["cpu_baseline"]["matrix_free_variant"]["value"]
            except Exception as e:       # noqa: BLE001 -- the GPU measurement above must not be lost with it
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # The legs below are informational: the headline line above is complete without them, and a failure in one of them
        # is reported in the line (never hidden, never allowed to lose the measurement already made).
        def informational(key, fn):
            try:
                out[key] = fn()
            except Exception as e:       # noqa: BLE001 -- recorded in the output
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}

        def host_leg():
            # the same batch through the host-buffer entry point (pageable numpy arrays in, H2D, one launch, D2H):
            # the PCIe-inclusive rate -- reported beside `value`, never as it
            bb.solve_host(pb, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)
            th = time.perf_counter()
            for _ in range(3):
                bb.solve_host(pb, num_iters=args.admm_iters, maxit=args.maxit, precision=args.precision)
            return 3 * B / (time.perf_counter() - th)
        if world == 1 and not args.no_latency:
            informational("p50_latency_ms_batch1", lambda: p50_latency(args.config, args.admm_iters))
            informational("host_buffers_solves_per_s", host_leg)
        if world == 1 and not args.no_kinodyn:
            informational("kinodyn_full_solve", lambda: kinodyn_leg(dev, args.kinodyn_batch, args.admm_iters, args.maxit, args.kinodyn_config,
                                                                    n_streams=args.kinodyn_streams))
            if args.kinodyn_config == "solo12_h20" and not args.kinodyn_main_only:   # BASELINE config 5's shape as well (1024 problems = its per-GPU share)
                informational("kinodyn_go2_h60", lambda: kinodyn_leg(dev, 1024, args.admm_iters, args.maxit, "go2_h60", n_streams=args.kinodyn_streams))
                informational("datagen_pass", lambda: datagen_leg(dev, args.kinodyn_batch, args.admm_iters))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The N > 1 path of bench.py on a one-GPU box: two fresh child processes started exactly as the driver starts them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ...`), both ranks on cuda:0 and gloo in
place of RCCL (BUNMPC_BENCH_ONE_DEVICE=1 -- the only difference from the driver's 8-GPU launch).  Checks that every leg is
rank-sharded (rank r owns problems [r B, (r + 1) B)), that telemetry reduces over the ranks and that the line keeps the
contract's fields."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(extra, nproc=2):
    env = dict(os.environ, BUNMPC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_rank_bench_shards_every_leg():
    from bunmpc_amd import batch as bb, problems
    B = 128
    out = _run_bench(["--steps", "2", "--warmup", "1", "--batch", str(B), "--config", "solo12_mixed", "--kinodyn-batch", "32", "--kinodyn-steps", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 2 * B
    assert "mixed" in out["metric"] and out["value"] > 0 and "cpu_baseline" not in out          # the CPU leg is N = 1 only
    assert abs(out["value"] - 2 * B * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]
    # the two ranks together solved exactly problems [0, 2 B): the summed FISTA iterations equal one process solving them all
    whole = bb.DeviceBatch(problems.make_batch("solo12_mixed", 2 * B), num_iters=10)
    whole.solve()
    st = whole.results()["stats"]
    assert out["fista_iters_per_solve"] == pytest.approx(st[:, 1:3].sum() / (2 * B), rel=1e-12)
    assert out["diverged"] == int((st[:, 5] != 0).sum())
    for leg in ("kinodyn_full_solve", "kinodyn_go2_h60"):
        k = out[leg]
        assert "error" not in k, k
        assert k["n_gpus"] == 2 and k["global_batch"] == 64 and k["value"] > 0 and k["roofline"]["frac"] > 0
        assert k["multi_stream"]["results_equal_single_stream"]
    # rank sharding of the whole-body batch: iteration statistics of the two shards = those of the first 64 problems
    import dataclasses
    from bunmpc_amd import urdf_model
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "solo12.json")).read())
    kb = KinoDynDeviceBatch(problems.make_wb_batch(model, 64), model, num_iters=10)
    kb.solve()
    r = kb.results()
    assert out["kinodyn_full_solve"]["ddp_iters_mean"] == pytest.approx(r["ik_iters"].mean(), rel=1e-12)
    assert out["kinodyn_full_solve"]["ddp_iters_max"] == int(r["ik_iters"].max())
    go2 = urdf_model.RobotModel.from_json(open(os.path.join(ROOT, "bunmpc_amd", "robots", "go2.json")).read())
    wb = problems.make_wb_batch(go2, 64, gait=dataclasses.replace(problems.TROT, nom_ht=0.30, gait_horizon=6.0), wb=problems.GO2_WB)
    kb = KinoDynDeviceBatch(wb, go2, num_iters=10)
    kb.solve()
    r = kb.results()
    assert out["kinodyn_go2_h60"]["ddp_not_converged"] == int((r["ik_status"] != 0).sum())
    assert out["kinodyn_go2_h60"]["ddp_iters_mean"] == pytest.approx(r["ik_iters"].mean(), rel=1e-12)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (torch.distributed.run,
    before it touches the GPU) and passes rank 0's line through.  Both ranks on cuda:0 over gloo here (one-GPU box)."""
    env = dict(os.environ, BUNMPC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--no-kinodyn"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["value"] > 0
    assert out["rccl"]["initialised"] and out["rccl"]["world"] == 2


def test_rccl_runs_at_world_size_one():
    """A plain `python bench.py` (N = 1, no launcher) brings RCCL up with one rank on the real device and sends every leg's
    barrier and MAX / SUM reductions through it: the collectives of the N > 1 path have executed on this code."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BUNMPC_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "256", "--no-cpu",
                        "--no-latency", "--kinodyn-batch", "64", "--kinodyn-steps", "1", "--kinodyn-main-only"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["rccl"]["initialised"], out["rccl"]
    assert out["rccl"]["backend"].startswith("nccl") and out["rccl"]["world"] == 1
    assert out["n_gpus"] == 1 and out["value"] > 0 and "error" not in out["kinodyn_full_solve"]


def test_kinodyn_workload_line_single_rank():
    """`--workload kinodyn`: the full KinoDynMP.optimize as the measured line, with its own roofline and cpu_baseline"""
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "kinodyn", "--kinodyn-config", "go2_h60", "--kinodyn-batch", "48",
                        "--steps", "1", "--warmup", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert "KinoDynMP" in out["metric"] and "H=60" in out["metric"] and out["dtype"] == "f64" and out["n_gpus"] == 1
    assert out["config"]["global_batch"] == 48 and out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    ks = out["details"]["ik_kernel_ms_per_solve"]
    assert all(ks[k] > 0 for k in ("ik_state_kernel", "ik_calcdiff_kernel", "ik_backward_kernel", "ik_forward_kernel"))
    assert np.isfinite(out["value"]) and out["value"] > 0

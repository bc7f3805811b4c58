"""world_size-2 gloo test of the multi-GPU path's host logic: each rank generates and owns its
slab of the batch, nothing is exchanged on the data path, telemetry reduces correctly and the
gathered slabs equal the single-process batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bunmpc_amd import problems, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(rank, world, per_rank)
    b = problems.make_batch("solo12_trot", per_rank, first=lo)
    # stand-in for the per-rank solve: a deterministic function of the slab (the GPU solve itself
    # is covered by the -m gpu tests; here only ownership / reduction / gather are under test)
    slab = b.x_init * 2.0 + b.dt[:, :9]
    t, c = sharding.reduce_telemetry(dist, torch, "cpu", 1.0 + rank, 10.0 * (rank + 1), [rank + 1, per_rank])
    full = sharding.gather_slabs(dist, torch, "cpu", slab)
    if rank == 0:
        np.savez(os.path.join(out_dir, "r.npz"), t=t, c=c, full=full, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world, per_rank = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    r = np.load(tmp_path / "r.npz")
    assert list(r["t"]) == [2.0, 20.0]                      # MAX over ranks
    assert list(r["c"]) == [3.0, 10.0]                      # SUM over ranks
    whole = problems.make_batch("solo12_trot", world * per_rank)
    assert np.array_equal(r["full"], whole.x_init * 2.0 + whole.dt[:, :9])
    # a rank's slab is exactly the corresponding slice of the global batch
    part = problems.make_batch("solo12_trot", per_rank, first=per_rank)
    for k in ("cnt_plan", "dt", "x_init", "X_nom", "X_ter"):
        assert np.array_equal(getattr(part, k), getattr(whole, k)[per_rank:])

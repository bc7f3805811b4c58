"""bunmpc_amd/pipeline.py: batches solved concurrently on separate HIP streams (one host thread each) give bit for bit the
results of solving them one after the other -- the C-ABI's per-call state is thread-local and the kernels share nothing."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROBOTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots")


def test_concurrent_batches_equal_sequential_ones():
    import torch
    from bunmpc_amd import problems, urdf_model
    from bunmpc_amd.kinodyn_batch import KinoDynDeviceBatch
    from bunmpc_amd.pipeline import StreamPool
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "solo12.json")).read())
    batches = [problems.make_wb_batch(model, B, seed=s) for B, s in ((96, 1), (40, 2), (130, 3), (7, 4))]
    seq = []
    for wb in batches:
        kb = KinoDynDeviceBatch(wb, model, device="cuda:0")
        kb.solve()
        seq.append(kb.results())
    kbs = [KinoDynDeviceBatch(wb, model, device="cuda:0") for wb in batches]
    pool = StreamPool("cuda:0", 3)
    pool.run([k.solve for k in kbs])
    for k, ref in zip(kbs, seq):
        got = k.results()
        for key in ("X", "F", "xs", "us", "ik_iters", "ik_status"):
            assert np.array_equal(got[key], ref[key]), key
    # an error in one job surfaces in the caller
    def boom():
        raise RuntimeError("job failed")
    with pytest.raises(RuntimeError, match="job failed"):
        pool.run([kbs[0].solve, boom])

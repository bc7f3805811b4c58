"""bmpc_perturb_batch_device (csrc/perturb.hip) against oracle/perturb_np.py on the same normal draws.  The oracle goes
through numpy's SVD-based pinv, the kernel through Gram-Schmidt: agreement is asked to 1e-9 (measured ~1e-13)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROBOTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bunmpc_amd", "robots")
FEET = ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"]
MU, SIGMA = [0.0, 0.01, -0.02, 0.03], [0.05, 0.1, 0.2, 0.2]


def _nominal(model, B, rng):
    from oracle import rbd_np
    q = np.zeros((B, 19))
    for b in range(B):
        base = rbd_np.neutral(model)
        base[:3] = [rng.normal(0, 1.0), rng.normal(0, 1.0), 0.24]
        base[7:] = np.array([0.0, 0.8, -1.6] * 4) + rng.normal(0, 0.1, 12)
        d = np.zeros(18)
        d[3:6] = rng.normal(0, 0.1, 3)
        q[b] = rbd_np.integrate(model, base, d)
        # lowest foot on the ground, so that some draws are rejected
        kin = rbd_np.Kin(model, q[b])
        q[b, 2] -= min(kin.frame_placement(n)[1][2] for n in FEET) - (0.0 if b % 2 else 0.01)
    return q, rng.normal(0, 0.5, (B, 18))


def test_sampler_matches_the_oracle_draw_for_draw():
    import torch
    from bunmpc_amd import perturbation, urdf_model
    from oracle import perturb_np
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "solo12.json")).read())
    rng = np.random.default_rng(8)
    B, K = 70, 5
    q, v = _nominal(model, B, rng)
    patterns = [[1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 1, 1], [0, 0, 0, 0], [1, 1, 0, 0], [0, 0, 0, 1]]
    cnt_plan = np.zeros((B, 3, 4, 4))                       # a contact plan: the flag is entry 0 of each foot's row
    for b in range(B):
        cnt_plan[b, 1, :, 0] = patterns[b % len(patterns)]
    z = rng.normal(size=(B, K, 36))
    s = perturbation.PerturbationSampler(model, FEET, MU, SIGMA, draws_per_call=K)
    dev = lambda a: torch.as_tensor(a, device="cuda:0")
    plan = dev(cnt_plan)
    qn, vn, ch = s.apply(dev(q), dev(v), plan[:, 1, :, 0], dev(z))
    qn, vn, ch = qn.cpu().numpy(), vn.cpu().numpy(), ch.cpu().numpy()
    n_rej = 0
    for b in range(B):
        rq, rv, k = perturb_np.sample(model, FEET, q[b], v[b], cnt_plan[b, 1, :, 0], z[b], MU, SIGMA)
        assert ch[b] == k, (b, ch[b], k)
        n_rej += k != 0
        if k < 0:
            assert np.all(qn[b] == q[b]) and np.all(vn[b] == v[b])
            continue
        if np.dot(rq[3:7], qn[b, 3:7]) < 0:
            rq[3:7] = -rq[3:7]                              # same rotation
        assert np.abs(qn[b] - rq).max() < 1e-9 and np.abs(vn[b] - rv).max() < 1e-9, (b, np.abs(qn[b] - rq).max(), np.abs(vn[b] - rv).max())
    assert n_rej > 5                                         # the rejection path was exercised


def test_sample_until_accepted_and_zero_sigma():
    import torch
    from bunmpc_amd import perturbation, urdf_model
    from oracle import rbd_np
    model = urdf_model.RobotModel.from_json(open(os.path.join(ROBOTS, "solo12.json")).read())
    rng = np.random.default_rng(9)
    B = 300
    q, v = _nominal(model, B, rng)
    contact = np.tile([1.0, 0.0, 0.0, 1.0], (B, 1))
    dev = lambda a: torch.as_tensor(a, device="cuda:0")
    s = perturbation.PerturbationSampler(model, FEET, [0, 0, 0, 0], SIGMA, draws_per_call=2)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    qn, vn, left = s.sample(dev(q), dev(v), dev(contact), generator=g)
    assert left.numel() == 0
    qn = qn.cpu().numpy()
    for b in range(0, B, 37):
        kin = rbd_np.Kin(model, qn[b])
        assert min(kin.frame_placement(n)[1][2] for n in FEET) >= 0.0
        assert abs(np.linalg.norm(qn[b, 3:7]) - 1) < 1e-14 and np.abs(qn[b] - q[b]).max() > 1e-4
    # sigma = 0: nothing moves (J * 0 has an empty row space: the velocity projector is the identity, applied to pos = 0)
    s0 = perturbation.PerturbationSampler(model, FEET, [0, 0, 0, 0], [0, 0, 0, 0], draws_per_call=1)
    qn, vn, ch = s0.apply(dev(q), dev(v), dev(contact), torch.zeros((B, 1, 36), dtype=torch.float64, device="cuda:0"))
    assert np.abs(qn.cpu().numpy() - q).max() < 1e-15 and torch.equal(vn.cpu(), torch.as_tensor(v))

"""Contact-conditioned perturbation of nominal states on the GPU (`bmpc_perturb_batch_device`, csrc/perturb.hip): the
sampler of `ISL/examples/iterative_algorithm/data_collection.py:188-262` for a whole batch of nominal states, so that
the initial conditions of the next batch of MPC solves never visit the host.  Draws come from torch's device generator
(Philox) unless given; the reference draws from numpy's global generator, so streams differ by construction while the
map from draws to states is the same (tests feed both the same draws).  No CPU fallback."""
import ctypes as C

from . import _lib
from .inverse_kinematics_cpp import as_device_model

# cfgs/data_collection_config.yaml:19-50 keys, in the order of bmpc_perturb_batch_t.mu / sigma
GROUPS = ("base_pos", "base_ori", "joint_pos", "vel")


class PerturbationSampler:
    def __init__(self, robot, eff_names, mu=(0.0, 0.0, 0.0, 0.0), sigma=(0.1, 0.7, 0.5, 0.2), draws_per_call=8, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("bunmpc_amd needs a GPU: there is no CPU fallback")
        self.torch, self.device = torch, torch.device(device)
        self.dev_model = as_device_model(robot)
        self.foot_frames = [self.dev_model.model.frame_id(n) for n in eff_names]
        self.mu, self.sigma, self.K = [float(x) for x in mu], [float(x) for x in sigma], int(draws_per_call)

    def apply(self, q, v, contact, z):
        """q (B,19), v (B,18), contact (B,4) strided view allowed, z (B,K,36): CUDA float64 tensors ->
        (q', v', chosen (B,) int32; -1 = every draw rejected, the row then holds the nominal state)"""
        torch = self.torch
        B, K = z.shape[0], z.shape[1]
        q, v, z = q.contiguous(), v.contiguous(), z.contiguous()
        q_out, v_out = torch.empty_like(q), torch.empty_like(v)
        chosen = torch.empty(B, dtype=torch.int32, device=q.device)
        d = _lib.PerturbBatch()
        d.B, d.K, d.model = B, K, self.dev_model.h
        d.foot_frame[:] = self.foot_frames
        d.mu[:] = self.mu
        d.sigma[:] = self.sigma
        d.q, d.v, d.z, d.contact = q.data_ptr(), v.data_ptr(), z.data_ptr(), contact.data_ptr()
        d.s_contact_b, d.s_contact_e = (contact.stride(0), contact.stride(1)) if B > 0 else (4, 1)
        d.q_out, d.v_out, d.chosen = q_out.data_ptr(), v_out.data_ptr(), chosen.data_ptr()
        _lib.check(_lib.lib().bmpc_perturb_batch_device(C.byref(d), C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)))
        return q_out, v_out, chosen

    def sample(self, q, v, contact, generator=None, max_rounds=16):
        """draw until every state has an accepted perturbation (the reference's `while min_ee_height >= 0`)"""
        torch = self.torch
        B = q.shape[0]
        q_out, v_out = q.clone(), v.clone()
        todo = torch.arange(B, device=q.device)
        for _ in range(max_rounds):
            if todo.numel() == 0:
                break
            z = torch.randn((todo.numel(), self.K, 36), dtype=torch.float64, device=q.device, generator=generator)
            qn, vn, ch = self.apply(q[todo], v[todo], contact[todo], z)
            ok = ch >= 0
            q_out[todo[ok]], v_out[todo[ok]] = qn[ok], vn[ok]
            todo = todo[~ok]
        return q_out, v_out, todo

"""Fused Adam over the model's flat parameter / gradient buffers (one kernel per step).

Same update as ``optim.Adam(model.parameters(), lr)`` at pretrain/roco_train.py:90 and
vqamed2019/train.py:160 (betas (0.9, 0.999), eps 1e-8, no weight decay).  Parameters whose
gradient stays zero (the reference's never-used ``norm2`` / ``fc``) do not move, which equals the
reference's "grad is None => skipped".
"""
from __future__ import annotations

import torch

from . import _lib as L


class FusedAdam:
    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps)]  # ReduceLROnPlateau-compatible
        self.m = torch.zeros_like(model.flat_params)
        self.v = torch.zeros_like(model.flat_params)
        self.step_count = 0
        self._for = model.flat_params.data_ptr()

    def zero_grad(self, set_to_none=False):
        self.model.flat_grads.zero_()

    def step(self, grad_scale=1.0, zero_grad=True):
        """p -= lr * m_hat / (sqrt(v_hat) + eps); grads are multiplied by grad_scale first
        (1/world_size under DDP) and zeroed in the same pass when zero_grad is set."""
        if self.model.flat_params.data_ptr() != self._for:
            raise L.MMVQAError("FusedAdam: the model was re-laid out (.to()/re-head) after the optimizer was built")
        g = self.param_groups[0]
        self.step_count += 1
        p = self.model.flat_params
        L.check(L.lib().mmvqa_adam(L.stream_ptr(), L.ptr(p), L.ptr(self.model.flat_grads), L.ptr(self.m),
                                   L.ptr(self.v), p.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                   self.step_count, grad_scale, 1 if zero_grad else 0))

    def state_dict(self):
        return dict(m=self.m, v=self.v, step=self.step_count, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        # in place: a scheduler built on this optimizer keeps pointing at the SAME list/dicts, so a later
        # ReduceLROnPlateau step still reaches mmvqa_adam after --resume
        for mine, theirs in zip(self.param_groups, sd["param_groups"]):
            mine.update(theirs)

"""Fused Adam over the model's flat parameter / gradient buffers (one kernel per step, or -- overlap_backward() -- one
per finished gradient range, enqueued while the backward pass is still running: the update is 32 bytes of HBM traffic
per parameter and no matrix work, the backbone backward is the opposite).

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
        self._stream = None          # overlap_backward(): the stream the early updates run on
        self._done = []              # [lo, hi) ranges already updated in the current step
        self._grad_scale = 1.0

    # ------------------------------------------------------------------ update beside the backward pass
    def overlap_backward(self, reducer=None, grad_scale=1.0):
        """Update a parameter range as soon as its gradient is final instead of after the whole backward pass.
        Single process: the model's gradient-ready announcements (Model.set_grad_ready_hook) start the range's Adam on a
        stream of its own, ordered behind the announcement's event.  With a ddp.GradReducer that exchanges gradients
        (world > 1 or the native RCCL route) the reducer owns the hook and calls back per bucket: the update is ordered
        behind that bucket's all-reduce (work.wait() puts the wait on the update's stream, not on the host).
        step() then only updates what was never announced and joins the streams.  The arithmetic per element is the same
        kernel with the same step number: results are bit-identical to the one-launch form (tests/test_hip_train.py).
        Contract: ONE backward per step (no gradient accumulation), zero_grad=True and this grad_scale in step()."""
        self._stream = torch.cuda.Stream()
        self._grad_scale = float(grad_scale)
        self._done = []
        if reducer is not None and reducer.exchanges():
            reducer.after_bucket = self._after_bucket
        else:
            self.model.set_grad_ready_hook(self._on_ready, with_event=True)

    def _adam(self, lo, hi, step, grad_scale, zero_grad, stream_ptr):
        g = self.param_groups[0]
        p = self.model.flat_params
        L.check(L.lib().mmvqa_adam(stream_ptr, L.ptr(p[lo:hi]), L.ptr(self.model.flat_grads[lo:hi]), L.ptr(self.m[lo:hi]),
                                   L.ptr(self.v[lo:hi]), hi - lo, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                   step, grad_scale, 1 if zero_grad else 0))

    def _early(self, lo, hi, ready=None, work=None, stream=None):
        if hi <= lo or any(a < hi and lo < b for a, b in self._done):
            return   # (a second backward before step(): this range already moved in this step)
        s = stream if stream is not None else self._stream
        if ready is not None:
            s.wait_event(ready)
        with torch.cuda.stream(s):
            if work is not None:
                work.wait()
            self._adam(lo, hi, self.step_count + 1, self._grad_scale, True, L.stream_ptr())
        if stream is not None:
            self._stream.wait_stream(stream)   # step() joins ONE stream
        self._done.append((lo, hi))

    def _on_ready(self, lo, hi, ready):
        self._early(lo, hi, ready=ready)

    def _after_bucket(self, lo, hi, work=None, stream=None):
        self._early(lo, hi, work=work, stream=stream)

    def zero_grad(self, set_to_none=False):
        self.model.flat_grads.zero_()

    def step(self, grad_scale=1.0, zero_grad=True):
        """p -= lr * m_hat / (sqrt(v_hat) + eps); grads are multiplied by grad_scale first
        (1/world_size under DDP) and zeroed in the same pass when zero_grad is set."""
        if self.model.flat_params.data_ptr() != self._for:
            raise L.MMVQAError("FusedAdam: the model was re-laid out (.to()/re-head) after the optimizer was built")
        self.step_count += 1
        n = self.model.flat_params.numel()
        if not self._done:
            self._adam(0, n, self.step_count, grad_scale, zero_grad, L.stream_ptr())
            return
        if not zero_grad or abs(grad_scale - self._grad_scale) > 1e-12 * abs(self._grad_scale):
            raise L.MMVQAError("FusedAdam.step: ranges were already updated during backward with zero_grad=True and "
                               f"grad_scale={self._grad_scale}; step(grad_scale={grad_scale}, zero_grad={zero_grad}) does not match")
        pos = 0
        for lo, hi in sorted(self._done):      # whatever backward never announced
            if lo > pos:
                self._adam(pos, lo, self.step_count, grad_scale, True, L.stream_ptr())
            pos = max(pos, hi)
        if pos < n:
            self._adam(pos, n, self.step_count, grad_scale, True, L.stream_ptr())
        torch.cuda.current_stream().wait_stream(self._stream)   # the next forward reads the updated parameters
        self._done = []

    def state_dict(self):
        return dict(m=self.m, v=self.v, step=self.step_count, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        # in place: a scheduler built on this optimizer keeps pointing at the SAME list/dicts, so a later
        # ReduceLROnPlateau step still reaches mmvqa_adam after --resume
        for mine, theirs in zip(self.param_groups, sd["param_groups"]):
            mine.update(theirs)

"""Losses of the MMBERT training loops as autograd Functions over the HIP kernels.

  mlm_loss      -- pretrain/roco_utils.py:235-236 (log_softmax + NLLLoss over ALL positions) and
                   :257-265 (argmax accuracy over target > 0), one fused pass over the logits
  asl_loss      -- models/asl_singlelabel.py:23-53
  supcon_loss   -- models/SupConLoss/loss.py:21-98 as called with features only (SimCLR)
  split_feat    -- models/SupConLoss/supcon_utils.py:259-261
"""
from __future__ import annotations

import torch

from . import _lib as L


def _padded(x2d):
    """[rows, V] view whose row stride is a multiple of 4 and base 16-byte aligned (copy if not)"""
    rows, V = x2d.shape
    if x2d.stride(1) == 1 and x2d.stride(0) % 4 == 0 and x2d.stride(0) >= V and x2d.data_ptr() % 16 == 0:
        return x2d, x2d.stride(0)
    ld = (V + 3) & ~3
    buf = torch.zeros(rows, ld, dtype=torch.float32, device=x2d.device)
    buf[:, :V] = x2d
    return buf[:, :V], ld


class _MLMLoss(torch.autograd.Function):
    """forward: one pass over the logits (row log-sum-exp, NLL, first-index argmax); backward: one streaming
    pass writing (softmax - onehot) * upstream / rows.  The upstream gradient is read on the device by the kernel:
    nothing is synchronised with the host between forward and backward."""

    @staticmethod
    def forward(ctx, logits, target):
        if not logits.is_cuda:
            raise L.MMVQAError("mlm_loss: GPU tensors only (no CPU fallback)")
        V = logits.shape[-1]
        x, ld = _padded(logits.reshape(-1, V))
        rows = x.shape[0]
        tgt = target.reshape(-1).contiguous().long()
        row_loss = torch.empty(rows, dtype=torch.float32, device=x.device)
        row_lse = torch.empty(rows, dtype=torch.float32, device=x.device)
        pred = torch.empty(rows, dtype=torch.int64, device=x.device)
        out3 = torch.empty(3, dtype=torch.float32, device=x.device)
        L.check(L.lib().mmvqa_mlm_loss(L.stream_ptr(), L.ptr(x), ld, L.ptr(tgt), L.ptr(row_loss), L.ptr(row_lse),
                                       L.ptr(pred), None, 0, None, 1.0, rows, V, L.ptr(out3)))
        ctx.save_for_backward(x, tgt, row_lse)
        ctx.ld, ctx.shape = ld, logits.shape
        ctx.mark_non_differentiable(pred, out3)
        return out3[0].clone(), pred.view(target.shape), out3

    @staticmethod
    def backward(ctx, gloss, _gp, _go):
        x, tgt, row_lse = ctx.saved_tensors
        rows, V = x.shape
        dld = (V + 3) & ~3
        dl = torch.empty(rows, dld, dtype=torch.float32, device=x.device)
        g = gloss.reshape(1).float().contiguous()
        L.check(L.lib().mmvqa_mlm_grad(L.stream_ptr(), L.ptr(x), ctx.ld, L.ptr(tgt), L.ptr(row_lse), L.ptr(dl), dld,
                                       L.ptr(g), 1.0 / rows, rows, V))
        return dl[:, :V].view(ctx.shape), None


def mlm_loss(logits, target):
    """returns (loss, pred[B,T] (argmax at every position), stats[3] = {loss, n_masked, n_correct})"""
    return _MLMLoss.apply(logits, target)


class _ASLLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, gamma_pos, gamma_neg, eps):
        if not logits.is_cuda:
            raise L.MMVQAError("asl_loss: GPU tensors only (no CPU fallback)")
        x, ld = _padded(logits)
        rows, Cc = x.shape
        tgt = target.contiguous().long()
        row_loss = torch.empty(rows, dtype=torch.float32, device=x.device)
        dl = torch.zeros(rows, (Cc + 3) & ~3, dtype=torch.float32, device=x.device)
        L.check(L.lib().mmvqa_asl_loss(L.stream_ptr(), L.ptr(x), ld, L.ptr(tgt), L.ptr(row_loss), L.ptr(dl),
                                       dl.stride(0), rows, Cc, gamma_pos, gamma_neg, eps, 1.0 / rows))
        ctx.save_for_backward(dl)
        ctx.C = Cc
        return row_loss.mean()

    @staticmethod
    def backward(ctx, gloss):
        (dl,) = ctx.saved_tensors
        return (dl * gloss)[:, :ctx.C], None, None, None, None


def asl_loss(logits, target, gamma_pos=0.0, gamma_neg=4.0, eps=0.1):
    return _ASLLoss.apply(logits, target, gamma_pos, gamma_neg, eps)


class _SupCon(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, temperature, base_temperature):
        if not features.is_cuda:
            raise L.MMVQAError("supcon_loss: GPU tensors only (no CPU fallback)")
        N, nv, D = features.shape
        if nv != 2:
            raise NotImplementedError("two views (supcon_utils.py:259-261)")
        f = torch.cat(torch.unbind(features, dim=1), dim=0).contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        df = torch.empty_like(f)
        ws = torch.empty(4 * N, dtype=torch.float32, device=f.device)
        L.check(L.lib().mmvqa_supcon_loss(L.stream_ptr(), L.ptr(f), L.ptr(loss), L.ptr(df), L.ptr(ws), N, D,
                                          temperature, base_temperature, 1.0))
        ctx.save_for_backward(df)
        ctx.N = N
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        (df,) = ctx.saved_tensors
        N = ctx.N
        g = df * gloss
        return torch.stack([g[:N], g[N:]], dim=1), None, None


def supcon_loss(features, temperature=0.07, base_temperature=0.07):
    return _SupCon.apply(features, temperature, base_temperature)


def split_feat(feat, bsz):
    f1, f2 = torch.split(feat, [bsz, bsz], dim=0)
    return torch.cat([f1.unsqueeze(1), f2.unsqueeze(1)], dim=1)

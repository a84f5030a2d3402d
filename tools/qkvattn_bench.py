#!/usr/bin/env python3
"""Isolated timing of the fused QKV projection + attention forward launch (csrc/qkvattn.hip) at the bench's shape
(B 16, T 32, 12 heads) beside the two launches it replaces (projection GEMM + attention), HIP-event timed."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hip_helpers import *  # noqa: E402,F401,F403


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B, T, heads, D = 16, 32, 12, 64
    H = heads * D
    xn = torch.randn(B * T, H, device=dev())
    W = torch.randn(3 * H, H, device=dev()) / math.sqrt(H)
    bias = torch.randn(3 * H, device=dev())
    mask = torch.ones(B, T, dtype=torch.long, device=dev())
    qkv = torch.zeros(B * T, 3 * H, device=dev())
    probs = torch.zeros(B, heads, T, T, device=dev())
    ctx = torch.zeros(B * T, H, device=dev())
    fused = timeit(lambda: L.check(L.lib().mmvqa_qkv_attention_fwd(L.stream_ptr(), P(xn), P(W), P(bias), P(mask), P(qkv), P(probs),
                                                                   P(ctx), B, T, H, heads, 0.3, 7)))
    d = L.GemmDesc()
    d.M, d.N, d.K = B * T, 3 * H, H
    d.A, d.a_ld, d.g_Cs = P(xn), H, H
    d.B, d.b_ld = P(W), H
    linear_geom(d)
    d.C, d.c_ld, d.bias = P(qkv), 3 * H, P(bias)
    best = min(timeit(lambda t=t: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_FWD, 0, t, L.stream_ptr()))) for t in (3, 5, 6))
    a = L.AttnDesc()
    a.q, a.k, a.v = P(qkv), P(qkv) + 4 * H, P(qkv) + 8 * H
    a.row_stride, a.head_stride = 3 * H, D
    a.out, a.out_row_stride, a.out_head_stride = P(ctx), H, D
    a.mask, a.mask_on_query, a.probs = P(mask), 0, P(probs)
    a.B, a.T, a.heads, a.sqrt_d, a.drop_p, a.seed = B, T, heads, math.sqrt(D), 0.3, 7
    att = timeit(lambda: L.check(L.lib().mmvqa_attention(C.byref(a), D, 0, L.stream_ptr())))
    fl = 2.0 * B * T * 3 * H * H + 4.0 * B * heads * T * T * D
    print(f"fused {fused:.1f} us ({fl / fused / 1e6:.1f} TFLOP/s) | projection GEMM (best of tiles 3/5/6) {best:.1f} us + attention {att:.1f} us = {best + att:.1f} us"
          f" | MMVQA_QA_DBG={os.environ.get('MMVQA_QA_DBG', '0')}")


if __name__ == "__main__":
    main()

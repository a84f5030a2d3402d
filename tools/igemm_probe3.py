#!/usr/bin/env python3
"""Probe 3: wgrad of the layer3 3x3 conv: effect of prologues, tiles and split-K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
from hip_helpers import *  # noqa

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

N, H, W, Cin, Cout, K, s, p = 16, 14, 14, 256, 256, 3, 1, 1
x = torch.randn(N * H * W, Cin, device=dev()); g = torch.randn(N * H * W, Cout, device=dev()); z = torch.randn(N * H * W, Cout, device=dev())
dw = torch.zeros(Cout, K * K * Cin, device=dev())
sc, sh = torch.rand(Cin, device=dev()), torch.rand(Cin, device=dev())
c3 = [torch.rand(Cout, device=dev()) for _ in range(3)]
fl = 2.0 * N * H * W * Cout * K * K * Cin
for mode in ("plain", "dz", "dz+bnrelu"):
    for tile in (1, 2, 3):
        for sk in (0, 1, 2, 4, 8):
            d = conv_desc_wgrad(g, x, N, H, W, Cin, Cout, K, s, p, dw)
            if mode != "plain": d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(c3[0]), P(c3[1]), P(c3[2])
            if mode == "dz+bnrelu": d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
            d.splitk = sk
            us = timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_WGRAD, 0, tile, L.stream_ptr())))
            print(f"{mode:10s} tile{tile} splitk={sk}: {us:7.1f}us {fl/us/1e6:5.1f} TF", flush=True)
# 1x1 conv wgrad (l3.conv1): M'=256 N'=1024 K'=3136
x1 = torch.randn(N * H * W, 1024, device=dev()); dw1 = torch.zeros(256, 1024, device=dev())
for tile in (1, 2, 3):
    for sk in (0, 1, 2, 4, 8):
        d = conv_desc_wgrad(g, x1, N, H, W, 1024, 256, 1, 1, 0, dw1); d.splitk = sk
        us = timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_WGRAD, 0, tile, L.stream_ptr())))
        print(f"1x1 plain  tile{tile} splitk={sk}: {us:7.1f}us {2.0*3136*256*1024/us/1e6:5.1f} TF", flush=True)

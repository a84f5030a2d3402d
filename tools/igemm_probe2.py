#!/usr/bin/env python3
"""Probe 2: plain GEMM throughput vs problem size and tile (calibrates the main loop against the chip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
from hip_helpers import *  # noqa

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 2048), (3136, 256, 2304), (12544, 256, 2304), (50176, 256, 2304), (3136, 1024, 2304)]:
    x = torch.randn(M, K, device=dev()); w = torch.randn(N, K, device=dev()) * 0.05
    z = torch.zeros(M, N, device=dev())
    out = []
    for tile in (1, 2, 3):
        d = L.GemmDesc(); d.M, d.N, d.K = M, N, K
        d.A, d.a_ld, d.g_Cs, d.B, d.b_ld = P(x), K, K, P(w), K
        linear_geom(d); d.C, d.c_ld = P(z), N
        us = timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_FWD, 0, tile, L.stream_ptr())))
        out.append(f"tile{tile}: {us:9.1f}us {2.0*M*N*K/us/1e6:6.1f} TF")
    print(f"M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(out), flush=True)

#!/usr/bin/env python3
"""Probe: time = a + b*K for one (M,N) with/without fused prologue/statistics (forward kind)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
from hip_helpers import *  # noqa

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

M, N = 3136, 256
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 3: M, N = int(sys.argv[2]), int(sys.argv[3])
print(f"M={M} N={N} tile={tile}")
for K in (64, 256, 1024, 2304, 4608):
    x = torch.randn(M, K, device=dev()); w = torch.randn(N, K, device=dev()) * 0.05
    z = torch.zeros(M, N, device=dev()); sc = torch.rand(K, device=dev()); sh = torch.rand(K, device=dev())
    stat = torch.zeros(16, N, 2, dtype=torch.float64, device=dev())
    res = []
    for mode in ("plain", "pro", "pro+stat"):
        d = L.GemmDesc(); d.M, d.N, d.K = M, N, K
        d.A, d.a_ld, d.g_Cs, d.B, d.b_ld = P(x), K, K, P(w), K
        linear_geom(d); d.C, d.c_ld = P(z), N
        if mode != "plain": d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
        if mode == "pro+stat": d.stat1 = P(stat)
        res.append(timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_FWD, 0, tile, L.stream_ptr()))))
    fl = 2.0 * M * N * K
    print(f"K={K:5d} fwd plain {res[0]:7.1f}us ({fl/res[0]/1e6:5.1f} TF)  +pro {res[1]:7.1f}us  +pro+stat {res[2]:7.1f}us")
d = L.GemmDesc(); d.M, d.N, d.K = 64, 64, 64
x = torch.randn(64, 64, device=dev()); z = torch.zeros(64, 64, device=dev())
d.A, d.a_ld, d.g_Cs, d.B, d.b_ld = P(x), 64, 64, P(x), 64; linear_geom(d); d.C, d.c_ld = P(z), 64
print("1-WG launch:", timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_FWD, 0, 3, L.stream_ptr()))), "us")

#!/usr/bin/env python3
"""Probe 4: the stem tap (M=200704, N=768, K=64): epilogue cost by mode / activation / tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
from hip_helpers import *  # noqa

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (M, Cc, HW) in ((200704, 64, 12544), (50176, 256, 3136)):
    N = 768
    f = torch.randn(M, Cc, device=dev()); w = torch.randn(N, Cc, device=dev()) * 0.1
    sc, sh = torch.rand(Cc, device=dev()), torch.rand(Cc, device=dev())
    v = torch.zeros(M // HW, N, device=dev()); dv = torch.randn(M // HW, N, device=dev())
    out = torch.zeros(M, N, device=dev())
    for tile in (1, 2, 3):
        res = []
        for mode in ("plain-store", "tapfwd-none", "tapfwd-relu", "tapfwd-serf", "tapbwd-serf"):
            d = L.GemmDesc(); d.M, d.N, d.K = M, N, Cc
            d.A, d.a_ld, d.g_Cs, d.B, d.b_ld = P(f), Cc, Cc, P(w), Cc
            linear_geom(d); d.C, d.c_ld = P(out), N
            d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
            if mode.startswith("tapfwd"):
                d.epi_mode, d.tap_HW, d.tap_out = L.EPI_TAP_FWD, HW, P(v)
                d.act = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "serf": L.ACT_SERF}[mode.split("-")[1]]
            elif mode.startswith("tapbwd"):
                d.epi_mode, d.tap_HW, d.tap_dv, d.act = L.EPI_TAP_BWD, HW, P(dv), L.ACT_SERF
            res.append(f"{mode} {timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), L.KIND_FWD, 0, tile, L.stream_ptr()))):8.1f}us")
        print(f"M={M} K={Cc} tile{tile}: " + " | ".join(res), flush=True)

#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel on the layer shapes of config 2 (resnet152 + transformer,
B=16): forward / dgrad / wgrad per shape and tile, HIP-event timed through the C ABI.
    python tools/igemm_bench.py [--tiles 0,1,2,3] [--filter l3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hip_helpers import *  # noqa: E402,F401,F403

B = 16
# name, N, H, W, Cin, Cout, K, stride, pad, count per step
SHAPES = [
    ("l1.conv1", B, 56, 56, 256, 64, 1, 1, 0, 2), ("l1.conv2", B, 56, 56, 64, 64, 3, 1, 1, 3),
    ("l1.conv3", B, 56, 56, 64, 256, 1, 1, 0, 3),
    ("l2.conv1", B, 28, 28, 512, 128, 1, 1, 0, 7), ("l2.conv2", B, 28, 28, 128, 128, 3, 1, 1, 7),
    ("l2.conv3", B, 28, 28, 128, 512, 1, 1, 0, 8), ("l2.0.conv2s2", B, 56, 56, 128, 128, 3, 2, 1, 1),
    ("l3.conv1", B, 14, 14, 1024, 256, 1, 1, 0, 35), ("l3.conv2", B, 14, 14, 256, 256, 3, 1, 1, 35),
    ("l3.conv3", B, 14, 14, 256, 1024, 1, 1, 0, 36),
    ("l4.conv1", B, 7, 7, 2048, 512, 1, 1, 0, 2), ("l4.conv2", B, 7, 7, 512, 512, 3, 1, 1, 2),
    ("l4.conv3", B, 7, 7, 512, 2048, 1, 1, 0, 3),
    ("tap.stem", B, 112, 112, 64, 768, 1, 1, 0, 1), ("tap.l1", B, 56, 56, 256, 768, 1, 1, 0, 1),
    ("tap.l3", B, 14, 14, 1024, 768, 1, 1, 0, 1),
    ("qkv", 1, 1, 512, 768, 2304, 1, 1, 0, 4), ("ffn1", 1, 1, 512, 768, 3072, 1, 1, 0, 4),
    ("ffn2", 1, 1, 512, 3072, 768, 1, 1, 0, 4), ("cls2", 1, 1, 512, 768, 30524, 1, 1, 0, 1),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--filter", default="")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--nopro", action="store_true", help="plain operand loads (no BN prologue): prices the fused prologues")
    ap.add_argument("--nostat", action="store_true", help="forward without the statistics epilogue")
    ap.add_argument("--persist", default="0", help="comma list of persistent-form workgroup counts to time beside 0 (= one workgroup per tile)")
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    persists = [int(t) for t in a.persist.split(",")]
    ws = torch.zeros(8 << 20, device=dev())
    tickets = torch.zeros(16384, dtype=torch.int32, device=dev())
    tot = {}
    print(f"{'shape':14s} {'kind':6s} tile {'M':>7s} {'N':>6s} {'K':>6s} {'us':>9s} {'TFLOP/s':>8s}")
    for name, N, H, W, Cin, Cout, K, s, p, cnt in SHAPES:
        if a.filter and a.filter not in name:
            continue
        OH, OW = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
        x = torch.randn(N * H * W, Cin, device=dev())
        w = torch.randn(Cout, K * K * Cin, device=dev()) * 0.05
        z = torch.zeros(N * OH * OW, Cout, device=dev())
        g = torch.randn(N * OH * OW, Cout, device=dev())
        dx = torch.zeros(N * H * W, Cin, device=dev())
        dw = torch.zeros(Cout, K * K * Cin, device=dev())
        sc, sh = torch.rand(Cin, device=dev()) + 0.5, torch.randn(Cin, device=dev()) * 0.1
        c3 = [torch.rand(Cout, device=dev()) for _ in range(3)]
        stat = torch.zeros(16, Cout, 2, dtype=torch.float64, device=dev())
        flops = 2.0 * N * OH * OW * Cout * K * K * Cin
        for kind in a.kinds.split(","):
          for persist in persists:
            for tile in tiles:
                if persist and tile not in (3, 5, 6):
                    continue
                if kind == "fwd":
                    d, _, _ = conv_desc_fwd(x, w, N, H, W, Cin, Cout, K, s, p, z)
                    if not a.nopro:
                        d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
                    if not a.nostat:
                        d.stat1 = P(stat)
                    kd = L.KIND_FWD
                elif kind == "dgrad":
                    d = conv_desc_dgrad(g, w, N, H, W, Cin, Cout, K, s, p, dx)
                    if not a.nopro:
                        d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(c3[0]), P(c3[1]), P(c3[2])
                    kd = L.KIND_DGRAD
                else:
                    d = conv_desc_wgrad(g, x, N, H, W, Cin, Cout, K, s, p, dw)
                    if not a.nopro:
                        d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(c3[0]), P(c3[1]), P(c3[2])
                        d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
                    if K > 1 and s == 1 and OH == H:
                        tab = torch.zeros(N * OH * OW, dtype=torch.int32, device=dev())
                        L.check(L.lib().mmvqa_pixmask(L.stream_ptr(), P(tab), N, OH, OW, H, W, K, K, s, p))
                        d.pixmask = P(tab)
                    kd = L.KIND_WGRAD
                if persist:
                    d.persist, d.splitk = persist, 1
                    d.sk_ws, d.sk_ws_floats, d.sk_cnt, d.sk_cnt_n = P(ws), ws.numel(), P(tickets), tickets.numel()
                us = timeit(lambda: L.check(L.lib().mmvqa_igemm(C.byref(d), kd, 0, tile, L.stream_ptr())))
                print(f"{name:14s} {kind:6s} {tile:4d} {d.M:7d} {d.N:6d} {d.K:6d} {us:9.1f} {flops / us / 1e6:8.1f}" + (f"  persist {persist}" if persist else ""), flush=True)
                if tile == tiles[0] and not persist:
                    tot[kind] = tot.get(kind, 0.0) + us * cnt
    print("weighted us per step (first tile option):", {k: round(v) for k, v in tot.items()}, "sum", round(sum(tot.values())))


if __name__ == "__main__":
    import ctypes as C
    main()

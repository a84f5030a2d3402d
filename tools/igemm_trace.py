#!/usr/bin/env python3
"""Per-workgroup phase timeline of the implicit-GEMM kernel (s_memtime stamps written by a -DIGEMM_TRACE build
of the library, built on the fly into tools/build/).  For every shape: span of the launch, the ramp of workgroup
start times, and the mean time a workgroup spends in setup / first tile / K loop / epilogue.
    python tools/igemm_trace.py --filter l3 --kinds fwd --tiles 3,5

Ablations: IGEMM_EXP="-DEXP_X" builds a variant library (results are wrong, timings are the point):
EXP_NOLOAD / EXP_NOSTORE / EXP_NOBAR (drop the K loop's global loads / LDS-write block / barrier), EXP_SAMETILE (every
workgroup loads workgroup (0,0)'s tiles), EXP_NOADVANCE (every K-tile re-reads the first), EXP_NOSTOREC (no output store)."""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build():
    src = os.path.join(ROOT, "mm-vqa_amd", "csrc")
    out = os.path.join(ROOT, "tools", "build")
    os.makedirs(out, exist_ok=True)
    exp = os.environ.get("IGEMM_EXP", "")
    lib = os.path.join(out, "libmmvqa_trace%s.so" % exp.replace("-D", "_").replace(" ", ""))
    srcs = [os.path.join(src, f) for f in ("igemm.hip", "attention.hip", "elementwise.hip", "augment.hip", "se.hip", "tapthin.hip", "qkvattn.hip", "engine.cpp",
                                           "abi.cpp")]
    if os.path.exists(lib) and all(os.path.getmtime(lib) > os.path.getmtime(s) for s in srcs):
        return lib
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
           "-DIGEMM_TRACE", *exp.split(), "-shared", "-x", "hip", "-I", os.path.join(ROOT, "include"), "-o", lib] + srcs
    subprocess.check_call(cmd)
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--filter", default="l3")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--build-only", action="store_true")
    a = ap.parse_args()
    lib = build()
    if a.build_only:
        return
    import mmvqa_amd._lib as L
    L.LIB_PATH = lib
    import torch
    import numpy as np
    import igemm_bench as IB
    from hip_helpers import conv_desc_fwd, conv_desc_dgrad, conv_desc_wgrad, P, dev
    Lh = L.lib()
    Lh.mmvqa_debug_set_trace.argtypes = [C.c_void_p]
    trace = torch.zeros((1 << 22) + (1 << 20), dtype=torch.int64, device=dev())
    assert Lh.mmvqa_debug_set_trace(trace.data_ptr()) == 0
    for name, N, H, W, Cin, Cout, K, s, p, cnt in IB.SHAPES:
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
        for kind in a.kinds.split(","):
            for tile in [int(t) for t in a.tiles.split(",")]:
                if kind == "fwd":
                    d, _, _ = conv_desc_fwd(x, w, N, H, W, Cin, Cout, K, s, p, z)
                    d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
                    d.stat1 = P(stat)
                    kd = L.KIND_FWD
                elif kind == "dgrad":
                    d = conv_desc_dgrad(g, w, N, H, W, Cin, Cout, K, s, p, dx)
                    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(c3[0]), P(c3[1]), P(c3[2])
                    kd = L.KIND_DGRAD
                else:
                    d = conv_desc_wgrad(g, x, N, H, W, Cin, Cout, K, s, p, dw)
                    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(c3[0]), P(c3[1]), P(c3[2])
                    d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
                    kd = L.KIND_WGRAD
                fn = lambda: L.check(Lh.mmvqa_igemm(C.byref(d), kd, 0, tile, L.stream_ptr()))
                us = IB.timeit(fn)
                trace.zero_()
                torch.cuda.synchronize()
                fn()
                torch.cuda.synchronize()
                tall = trace.cpu().numpy()
                t = tall[:1 << 22].reshape(-1, 8)
                stl = tall[1 << 22:].reshape(-1, 8)
                sel = t[:, 0] != 0
                stl = stl[:int(sel.sum())]
                t = t[sel]
                nwg = len(t)
                r0 = t[:, 6].min()
                span = (t[:, 7].max() - r0) * 0.01          # us (100 MHz)
                st = np.sort(t[:, 6] - r0) * 0.01
                dur_real = (t[:, 7] - t[:, 6]) * 0.01
                dur_tick = (t[:, 4] - t[:, 0]).astype(np.float64)
                tick = dur_real.sum() / dur_tick.sum()       # us per s_memtime tick
                ph = [(t[:, i + 1] - t[:, i]).mean() * tick for i in range(4)]
                life = dur_real
                xcc = (t[:, 5] >> 32) & 0xF
                cu = (t[:, 5] >> 8) & 0xF
                se = (t[:, 5] >> 13) & 0x7
                ncu = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
                print(f"{name:12s} {kind:6s} tile {tile} M={d.M} N={d.N} K={d.K}  {us:7.1f} us  wgs {nwg}  "
                      f"start p50/p90/max {st[nwg // 2]:5.1f}/{st[int(nwg * .9)]:5.1f}/{st[-1]:5.1f}  "
                      f"phases setup {ph[0]:5.2f} first {ph[1]:5.2f} loop {ph[2]:5.2f} epi {ph[3]:5.2f}  "
                      f"life mean {life.mean():5.1f} max {life.max():5.1f}  span {span:5.1f} us  CUs {ncu}  MHz {1 / tick:5.0f}  "
                      f"stall vm {stl[:, 0].mean() * tick:5.2f} lgkm {stl[:, 1].mean() * tick:5.2f} bar {stl[:, 2].mean() * tick:5.2f}  "
                      f"epi: stage {(stl[:, 4] - t[:, 3]).mean() * tick:5.2f} pre {(stl[:, 7] - stl[:, 4]).mean() * tick:5.2f} passes {(stl[:, 5] - stl[:, 7]).mean() * tick:5.2f} "
                      f"red {(stl[:, 6] - stl[:, 5]).mean() * tick:5.2f} final {(t[:, 4] - stl[:, 6]).mean() * tick:5.2f}", flush=True)


if __name__ == "__main__":
    main()

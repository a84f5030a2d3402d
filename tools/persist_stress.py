#!/usr/bin/env python3
"""Stress of the persistent (stream-K) GEMM form's hand-off: layer-sized forward / data-gradient products,
persistent vs one-workgroup-per-tile results compared element by element over many repetitions, with and without a second
stream keeping the chip busy (uneven load is where a broken hand-off shows: MI355X_MICROARCH.md).  Prints the worst
relative difference per shape and the number of repetitions that differed by more than 1e-5."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hip_helpers import *  # noqa: E402,F401,F403

SHAPES = [("l3.conv2", 16, 14, 14, 256, 256, 3, 1, 1), ("l3.conv1", 16, 14, 14, 1024, 256, 1, 1, 0),
          ("l3.0.conv2s2", 16, 28, 28, 256, 256, 3, 2, 1), ("l2.conv2", 16, 28, 28, 128, 128, 3, 1, 1),
          ("qkv", 1, 1, 512, 768, 2304, 1, 1, 0)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ws = torch.zeros(8 << 20, device=dev())
    tickets = torch.zeros(16384, dtype=torch.int32, device=dev())
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=dev())
    for name, N, H, W, Cin, Cout, K, s, p in SHAPES:
        OH, OW = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
        x = torch.randn(N * H * W, Cin, device=dev())
        w = torch.randn(Cout, K * K * Cin, device=dev()) * 0.05
        g = torch.randn(N * OH * OW, Cout, device=dev())
        z = torch.randn(N * OH * OW, Cout, device=dev())
        sc, sh = torch.rand(Cin, device=dev()) + 0.5, torch.randn(Cin, device=dev()) * 0.1
        c3 = [torch.rand(Cout, device=dev()) for _ in range(3)]
        for kind in ("fwd", "dgrad"):
            for tile in (3, 5, 6):
                for G in (256, 512):
                    for busy in (False, True):
                        worst, bad = 0.0, 0
                        for r in range(reps):
                            outs = []
                            for persist in (0, G):
                                if kind == "fwd":
                                    out = torch.zeros(N * OH * OW, Cout, device=dev())
                                    d, _, _ = conv_desc_fwd(x, w, N, H, W, Cin, Cout, K, s, p, out)
                                    d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
                                    kd = L.KIND_FWD
                                else:
                                    out = torch.zeros(N * H * W, Cin, device=dev())
                                    d = conv_desc_dgrad(g, w, N, H, W, Cin, Cout, K, s, p, out)
                                    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(c3[0]), P(c3[1]), P(c3[2])
                                    kd = L.KIND_DGRAD
                                d.persist, d.splitk = persist, 1
                                d.sk_ws, d.sk_ws_floats, d.sk_cnt, d.sk_cnt_n = P(ws), ws.numel(), P(tickets), tickets.numel()
                                if busy and persist:
                                    with torch.cuda.stream(side):
                                        for _ in range(2):
                                            big @ big
                                L.check(L.lib().mmvqa_igemm(C.byref(d), kd, 0, tile, L.stream_ptr()))
                                outs.append(out)
                            torch.cuda.synchronize()
                            e = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max())
                            worst = max(worst, e)
                            bad += e > 1e-5
                        if bad or worst > 1e-5:
                            print(f"{name:13s} {kind:5s} tile {tile} G {G} busy {busy}: worst {worst:.2e}, {bad}/{reps} repetitions differ", flush=True)
        print(f"{name}: done (tickets sum {int(tickets.abs().sum())})", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Times the squeeze-excite fully connected kernels (csrc/se.hip) on the tf_efficientnetv2_m shapes, through the C ABI.
    python tools/se_bench.py [--batch 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hip_helpers import *  # noqa: E402,F401,F403

SHAPES = [("stage3", 640, 40, 7), ("stage4", 1056, 44, 14), ("stage5", 1824, 76, 18), ("stage6", 3072, 128, 5)]


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    B = ap.parse_args().batch
    tf = tb = 0.0
    for name, mid, rd, cnt in SHAPES:
        g = lambda *s: torch.randn(*s, device=dev())  # noqa: E731
        pool, Wr, br, We, be = g(B, mid), g(rd, mid) / mid ** 0.5, g(rd), g(mid, rd) / rd ** 0.5, g(mid)
        rpre, r, gpre, gate = g(B, rd), g(B, rd), g(B, mid), g(B, mid)
        dgate, dpool = g(B, mid), g(B, mid)
        dWe, dbe, dWr, dbr = torch.zeros_like(We), torch.zeros_like(be), torch.zeros_like(Wr), torch.zeros_like(br)
        scratch = torch.zeros(L.lib().mmvqa_se_fc_bwd_scratch_floats(B, mid, rd), device=dev())
        f = timeit(lambda: L.check(L.lib().mmvqa_se_fc_fwd(L.stream_ptr(), P(pool), P(Wr), P(br), P(We), P(be), P(rpre), P(r),
                                                           P(gpre), P(gate), B, mid, rd)))
        b = timeit(lambda: L.check(L.lib().mmvqa_se_fc_bwd(L.stream_ptr(), P(dgate), P(gpre), P(r), P(rpre), P(pool), P(We),
                                                           P(Wr), P(dWe), P(dbe), P(dWr), P(dbr), P(dpool), P(scratch), B, mid, rd)))
        print(f"{name} mid {mid:5d} rd {rd:4d} x{cnt:2d}: fwd (2 launches) {f:6.1f} us   bwd (memset + 2 launches) {b:6.1f} us")
        tf += f * cnt; tb += b * cnt
    print(f"per config-3 step: fwd {tf / 1e3:.2f} ms, bwd {tb / 1e3:.2f} ms")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Block-end launch (relu(bn3(z) + idn) with both BatchNorm folds, csrc/elementwise.hip) over the four ResNet-152 layer
shapes at batch 16: time per launch and fraction of 8 TB/s for the workgroup targets MMVQA_BAR_WGS selects (read at the
first launch: one process per variant).   python tools/blockend_bench.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("layer1", 16 * 56 * 56, 256, 3), ("layer2", 16 * 28 * 28, 512, 8), ("layer3", 16 * 14 * 14, 1024, 36), ("layer4", 16 * 7 * 7, 2048, 3)]


def one():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from mmvqa_amd import _lib as L
    from test_hip_ops import _fold, _spread
    tot = 0.0
    line = []
    for name, rows, Cc, per_step in SHAPES:
        z, idn, out = (torch.randn(rows, Cc, device="cuda") for _ in range(3))
        sums = torch.stack([z.double().sum(0).cpu(), (z.double() ** 2).sum(0).cpu()], 1)
        st = _spread(sums, 4)
        g, b = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        outs = [torch.zeros(Cc, device="cuda") for _ in range(4)]
        f = _fold(st, 4, 0, 1, rows, g, beta=b, out0=outs[0], out1=outs[1], out2=outs[2], out3=outs[3])
        go = lambda: L.check(L.lib().mmvqa_bn_add_relu_fold(L.stream_ptr(), L.ptr(z), C.byref(f), L.ptr(idn), None, L.ptr(out), rows, Cc))
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us * per_step
        line.append(f"{name} {us:.1f} us ({rows * Cc * 12 / us / 1e6 / 8:.2f})")
    print(f"target {os.environ.get('MMVQA_BAR_WGS', 'default')}: " + ", ".join(line) + f"; per step {tot / 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for w in ("512", "1024", "2048", "4096", "8192", "16384", "65536"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env={**os.environ, "MMVQA_BAR_WGS": w})

#!/usr/bin/env python3
"""Adam pass over a config-2-sized flat buffer (133 M parameters, 32 B of HBM traffic each): time per launch and
fraction of 8 TB/s, for the launch shapes MMVQA_ADAM_UNROLL / MMVQA_ADAM_WGS select (read at the first launch: one
process per variant).   python tools/adam_bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    import torch
    from mmvqa_amd import _lib as L
    n = 133_000_000 // 4 * 4
    p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4))
    v.abs_()

    def go(k):
        L.check(L.lib().mmvqa_adam(L.stream_ptr(), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, 2e-5, 0.9, 0.999, 1e-8, k, 1.0, 1))
    for k in range(3):
        go(k + 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(10):
        go(k + 4)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"unroll {os.environ.get('MMVQA_ADAM_UNROLL', '2')} wgs {os.environ.get('MMVQA_ADAM_WGS', '4096')}: {ms:.3f} ms, "
          f"{n * 32 / ms / 1e9:.2f} TB/s = {n * 32 / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for u in ("1", "2", "3", "4"):
            for w in ("2048", "4096", "8192", "16384"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env={**os.environ, "MMVQA_ADAM_UNROLL": u, "MMVQA_ADAM_WGS": w})

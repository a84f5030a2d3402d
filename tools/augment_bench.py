#!/usr/bin/env python3
"""Throughput of the device input pipeline (mmvqa_amd.augment) on a batch of synthetic decoded images already copied
to HBM: images/s of the train and validation chains, and the achieved byte rate against the algorithmic bytes
(input read once + every byte-valued stage's read/write + the fp32 output), HIP-event timed.
    python tools/augment_bench.py [--batch 64] [--h 500] [--w 700]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mmvqa_amd import augment as AU  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--h", type=int, default=500)
    ap.add_argument("--w", type=int, default=700)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (a.h, a.w, 3)).astype(np.uint8) for _ in range(a.batch)]
    S = 224
    out = {}
    for train in (False, True):
        aug = AU.DeviceAugment(train=train)
        params = AU.sample_params(a.batch, generator=torch.Generator().manual_seed(0)) if train else None
        for _ in range(3):
            aug(imgs, params=params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            aug(imgs, params=params)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters
        # device time alone: replay the enqueued work with events around one call (host part overlaps the GPU part)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        aug(imgs, params=params)
        e1.record()
        torch.cuda.synchronize()
        ev = e0.elapsed_time(e1) * 1e-3
        px = S * S * 3
        rw, rh = AU.resized_size(a.w, a.h, S)
        tmp_rows = a.h if rh >= a.h else a.h           # (upper bound: every source row feeds the vertical pass)
        stage1 = a.h * a.w * 3 + 2 * tmp_rows * S * 3 + px          # read source, write+read the horizontal pass, write a0
        if train:
            byts = stage1 + (px + 2 * S * S * 3 + px) + 2 * px + 4 * 3 * px + px + 4 * px
        else:
            byts = stage1 + px + 4 * px
        out["train" if train else "val"] = dict(images_per_s_wall=a.batch / wall, ms_per_batch_wall=wall * 1e3,
                                                ms_per_batch_events=ev * 1e3, algorithmic_mb_per_image=byts / 1e6,
                                                gbps_events=byts * a.batch / ev / 1e9)
    out["config"] = dict(batch=a.batch, h=a.h, w=a.w, note="includes the host-side coefficient tables and the H2D copy of the batch in the wall figure")
    print(json.dumps(out))


if __name__ == "__main__":
    main()

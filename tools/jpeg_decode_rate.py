#!/usr/bin/env python3
"""What the JPEG stage of the input pipeline costs on the host, against what one GPU consumes.

The reference decodes on the host: `Image.open(path).convert('RGB')` in DataLoader workers (pretrain/roco_utils.py:573-587,
vqamed2019/utils.py:246, models/SupConLoss/supcon_utils.py:223; 16 workers in README.md:18).  This build moves everything
AFTER the decode to the device (mmvqa_amd.augment); the decode itself stays on the host.  This tool measures whether that
is the bottleneck of the step on THIS box:

  * decode rate per core: Pillow (libjpeg-turbo) `Image.open(BytesIO(jpeg)).convert('RGB')` on 500x700 baseline JPEGs
    (the ROCO / VQA-Med image scale), one process pinned to one core;
  * decode rate of a process pool over the cores this process may use (what a DataLoader with that many workers gets);
  * the hand-over: decoded uint8 batches in pinned memory -> HBM, alone and overlapped with GPU work on another stream;
  * the need: images/s one GPU consumes at the measured step time (batch 16 per step).

    python tools/jpeg_decode_rate.py [--images 256] [--h 500] [--w 700] [--step_ms 23.5] > profiles/round3_jpeg_decode.json
"""
import argparse
import io
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def synth_jpegs(n, h, w, quality=90, seed=0):
    """medical-image-like content: smooth structures + texture + text-like edges (so that the entropy-coded size per
    pixel is realistic: ~0.5-1 bit/pixel/channel at quality 90), baseline (non-progressive) JPEG as cameras/PACS export"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = []
    for i in range(n):
        img = np.zeros((h, w), np.float32)
        for _ in range(6):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(30, 220)
            img += rng.uniform(40, 160) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))
        img += rng.normal(0, 6, (h, w))
        img[rng.integers(0, h - 20):, : rng.integers(40, 200)] *= 0.3
        rgb = np.stack([img, img * rng.uniform(0.9, 1.0), img * rng.uniform(0.85, 1.0)], -1).clip(0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="JPEG", quality=quality)
        out.append(buf.getvalue())
    return out


def decode_all(jpegs):
    n = 0
    for b in jpegs:
        im = Image.open(io.BytesIO(b)).convert("RGB")     # pretrain/roco_utils.py:575
        n += im.size[0]
    return n


def _worker(args):
    jpegs, reps = args
    t0 = time.perf_counter()
    for _ in range(reps):
        decode_all(jpegs)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--h", type=int, default=500)
    ap.add_argument("--w", type=int, default=700)
    ap.add_argument("--step_ms", type=float, default=23.5, help="measured GPU step time at batch 16 (bench.py)")
    a = ap.parse_args()
    jpegs = synth_jpegs(a.images, a.h, a.w)
    kb = sum(len(b) for b in jpegs) / len(jpegs) / 1e3
    out = dict(config=dict(images=a.images, h=a.h, w=a.w, jpeg_kb_mean=kb, quality=90, pillow=Image.__version__ if hasattr(Image, "__version__") else None))
    # ---- one core
    decode_all(jpegs[:8])
    t0 = time.perf_counter()
    decode_all(jpegs)
    one = a.images / (time.perf_counter() - t0)
    out["decode_images_per_s_one_core"] = one
    # ---- a pool over the usable cores
    cores = usable_cores()
    reps = 2
    with mp.get_context("fork").Pool(cores) as pool:
        chunks = [(jpegs, reps)] * cores
        pool.map(_worker, [(jpegs[:4], 1)] * cores)            # warm the workers
        t0 = time.perf_counter()
        pool.map(_worker, chunks)
        wall = time.perf_counter() - t0
    out["decode_images_per_s_pool"] = dict(workers=cores, images_per_s=cores * reps * a.images / wall)
    need = 16.0 / (a.step_ms * 1e-3)
    out["need_images_per_s_per_gpu"] = dict(step_ms=a.step_ms, batch=16, images_per_s=need,
                                            cores_needed_per_gpu=need / one, eight_gpus=8 * need,
                                            cores_needed_for_eight_gpus=8 * need / one)
    # ---- hand-over of decoded bytes: pinned host memory -> HBM, alone and beside GPU work
    try:
        import torch
        if torch.cuda.is_available():
            B = 64
            host = torch.empty((B, a.h, a.w, 3), dtype=torch.uint8).pin_memory()
            dev = torch.empty_like(host, device="cuda")
            copy_stream = torch.cuda.Stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):
                dev.copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                dev.copy_(host, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            gb = host.numel() / 1e9
            res = dict(batch=B, mb=host.numel() / 1e6, ms_alone=ms, gbps_alone=gb / (ms * 1e-3), images_per_s_alone=B / (ms * 1e-3))
            # beside a compute stream that keeps the chip busy (a large fp32 matmul loop stands in for the training step)
            x = torch.randn(4096, 4096, device="cuda")
            for _ in range(3):
                y = x @ x            # (first call initialises the GEMM library)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                y = x @ x
            torch.cuda.synchronize()
            compute_alone = time.perf_counter() - t0
            t0 = time.perf_counter()
            with torch.cuda.stream(copy_stream):
                for _ in range(10):
                    dev.copy_(host, non_blocking=True)
            for _ in range(20):
                y = x @ x
            torch.cuda.synchronize()
            both = time.perf_counter() - t0
            res.update(compute_alone_ms=compute_alone * 1e3, compute_plus_10_copies_ms=both * 1e3,
                       ten_copies_alone_ms=10 * ms,
                       note="copies on their own stream beside compute: wall = max, not sum, when the hand-over is hidden")
            out["h2d_decoded_uint8"] = res
            del y
    except Exception as ex:   # the host figures stand on their own
        out["h2d_decoded_uint8"] = dict(error=repr(ex))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

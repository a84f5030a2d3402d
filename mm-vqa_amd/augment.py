"""Device input pipeline: the reference's image transforms on the GPU.

Reference (host, per sample, torchvision transforms on PIL images):
  pretrain/roco_train.py:98-112   Resize(224) CenterCrop(224) RandomResizedCrop(224, (0.95,1.05), (0.95,1.05))
                                  RandomRotation(5) ColorJitter(0.05 x4) ToTensor Normalize(0.5, 0.5)
  vqamed2019/train.py:179-200     same chain with scale/ratio (0.75,1.25), RandomRotation(10), ColorJitter(0.4 x4)
  validation / test               Resize(224) CenterCrop(224) ToTensor Normalize
Here: decoded uint8 RGB images (JPEG decoding stays on the host: the image has no GPU JPEG decoder) are copied to
HBM once and every transform runs as a HIP kernel (csrc/augment.hip) that restates Pillow's byte arithmetic bit for
bit; the random parameters are drawn on the host from torch's generator in torchvision's order (torchvision is a
third-party dependency, unpinned by the reference and absent from the build image: its thin parameter-sampling
wrappers are restated from the published source -- "parity unpinned" for the draw order; the pixel arithmetic is
pinned against Pillow itself, tests/test_augment.py).

    aug = DeviceAugment(train=True)                      # ROCO pre-training settings
    x = aug(list_of_uint8_HWC_arrays)                    # -> float32 [B, 3, 224, 224] on the GPU
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


# --------------------------------------------------------------------------- geometry helpers (torchvision semantics)
def resized_size(w, h, size):
    """transforms.Resize(int): the shorter side becomes `size`, the other int(size * long / short)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def center_crop_offset(w, h, size):
    """transforms.CenterCrop: int(round((dim - size) / 2.0)) (Python's round-half-to-even)"""
    return int(round((w - size) / 2.0)), int(round((h - size) / 2.0))


def sample_params(n, size=224, scale=(0.95, 1.05), ratio=(0.95, 1.05), degrees=5.0,
                  jitter=(0.05, 0.05, 0.05, 0.05), generator=None):
    """Per-image random parameters in the order torchvision's Compose draws them:
    RandomResizedCrop.get_params -> RandomRotation.get_params -> ColorJitter.get_params."""
    g = generator

    def uni(a, b):
        return torch.empty(1).uniform_(a, b, generator=g).item()

    out = []
    for _ in range(n):
        height = width = size
        area = height * width
        log_ratio = torch.log(torch.tensor(ratio))
        box = None
        for _try in range(10):
            target_area = area * uni(scale[0], scale[1])
            aspect = math.exp(uni(float(log_ratio[0]), float(log_ratio[1])))
            w = int(round(math.sqrt(target_area * aspect)))
            h = int(round(math.sqrt(target_area / aspect)))
            if 0 < w <= width and 0 < h <= height:
                i = int(torch.randint(0, height - h + 1, size=(1,), generator=g).item())
                j = int(torch.randint(0, width - w + 1, size=(1,), generator=g).item())
                box = (i, j, h, w)
                break
        if box is None:   # fallback: central crop
            in_ratio = float(width) / float(height)
            if in_ratio < min(ratio):
                w = width
                h = int(round(w / min(ratio)))
            elif in_ratio > max(ratio):
                h = height
                w = int(round(h * max(ratio)))
            else:
                w, h = width, height
            box = ((height - h) // 2, (width - w) // 2, h, w)
        angle = float(uni(-float(degrees), float(degrees)))
        order = torch.randperm(4, generator=g).tolist()
        b = float(uni(max(0.0, 1 - jitter[0]), 1 + jitter[0]))
        c = float(uni(max(0.0, 1 - jitter[1]), 1 + jitter[1]))
        s = float(uni(max(0.0, 1 - jitter[2]), 1 + jitter[2]))
        hh = float(uni(-jitter[3], jitter[3]))
        out.append(dict(box=box, angle=angle, order=order, brightness=b, contrast=c, saturation=s, hue=hh))
    return out


def hue_shift_u8(hue_factor):
    """torchvision F_pil.adjust_hue: np_h += np.array(hue_factor * 255).astype("uint8") (uint8 wrap-around)"""
    with np.errstate(invalid="ignore"):
        return int(np.array(np.int64(hue_factor * 255)).astype("uint8"))


def rotate_fix(angle, w, h):
    """Image.rotate's matrix (PIL/Image.py) turned into the 16.16 coefficients of Geometry.c affine_fixed"""
    angle = angle % 360.0
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    if angle == 0:
        m = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]      # Image.rotate returns a copy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(v) if v >= 0 else int(math.floor(v))

    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
            fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def coeffs(in_size, in0, in1, out_size):
    """(bounds [out][2] int32, kk [out][ksize] int32, ksize) from the library's HOST routine"""
    lib = L.lib()
    ks = lib.mmvqa_resample_coeffs(in_size, float(in0), float(in1), out_size, None, None, 0)
    if ks <= 0:
        L.check(ks)
    b = np.zeros((out_size, 2), np.int32)
    k = np.zeros((out_size, ks), np.int32)
    r = lib.mmvqa_resample_coeffs(in_size, float(in0), float(in1), out_size, b.ctypes.data_as(C.c_void_p),
                                  k.ctypes.data_as(C.c_void_p), ks)
    if r != ks:
        L.check(r if r < 0 else -1)
    return b, k, ks


class _Pack:
    """host-side builder of one int32 table blob + job array, uploaded with two copies per stage"""

    def __init__(self):
        self.tabs, self.n = [], 0

    def add(self, arr):
        a = np.ascontiguousarray(arr, np.int32).reshape(-1)
        off = self.n
        self.tabs.append(a)
        self.n += a.size
        return off

    def upload(self, dev):
        blob = np.concatenate(self.tabs) if self.tabs else np.zeros(1, np.int32)
        return torch.from_numpy(blob).to(dev)


class DeviceAugment:
    def __init__(self, size=224, train=True, scale=(0.95, 1.05), ratio=(0.95, 1.05), degrees=5.0,
                 jitter=(0.05, 0.05, 0.05, 0.05), mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), device="cuda"):
        self.size, self.train = int(size), bool(train)
        self.scale, self.ratio, self.degrees, self.jitter = tuple(scale), tuple(ratio), float(degrees), tuple(jitter)
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self.dev = torch.device(device)

    # ---- one resample stage: list of (src_ptr, sh, sw, box(x,y,w,h), (rw,rh), (ox,oy)) -> uint8 [B,S,S,3]
    def _resample(self, specs):
        S, dev, lib = self.size, self.dev, L.lib()
        B = len(specs)
        pack, metas = _Pack(), []
        cache = {}
        for (_src, _sh, _sw, (bx, by, bw, bh), (rw, rh), (ox, oy)) in specs:
            kh = ("h", bw, rw)
            if kh not in cache:
                b, k, ks = coeffs(bw, 0, bw, rw)
                cache[kh] = (pack.add(b), pack.add(k), ks, b)
            kv = ("v", bh, rh)
            if kv not in cache:
                b, k, ks = coeffs(bh, 0, bh, rh)
                cache[kv] = (pack.add(b), pack.add(k), ks, b)
            vb = cache[kv][3][oy:oy + S]
            ty0 = int(vb[:, 0].min())
            tyn = int((vb[:, 0] + vb[:, 1]).max()) - ty0
            metas.append((cache[kh], cache[kv], ty0, tyn))
        tabs = pack.upload(dev)
        max_rows = max(m[3] for m in metas)
        tmp = torch.empty(B, max_rows, S, 3, dtype=torch.uint8, device=dev)
        dst = torch.empty(B, S, S, 3, dtype=torch.uint8, device=dev)
        jobs = (L.ResampleJob * B)()
        base = tabs.data_ptr()
        for n, ((src, sh, sw, (bx, by, bw, bh), (rw, rh), (ox, oy)), (ch, cv, ty0, tyn)) in enumerate(zip(specs, metas)):
            j = jobs[n]
            j.src, j.sh, j.sw, j.spitch = src, sh, sw, sw * 3
            j.bx, j.by, j.bw, j.bh, j.rw, j.rh, j.ox, j.oy, j.ty0, j.tyn = bx, by, bw, bh, rw, rh, ox, oy, ty0, tyn
            j.tmp = tmp[n].data_ptr()
            j.dst, j.dpitch = dst[n].data_ptr(), S * 3
            j.hb, j.hk, j.hks = base + 4 * ch[0], base + 4 * ch[1], ch[2]
            j.vb, j.vk, j.vks = base + 4 * cv[0], base + 4 * cv[1], cv[2]
        jobs_dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        L.check(lib.mmvqa_aug_resample(L.stream_ptr(), L.ptr(jobs_dev), B, max_rows, S, S))
        self._keep = (tabs, tmp, jobs_dev)     # alive until the stream has consumed them (next call replaces them)
        return dst

    def __call__(self, images, params=None, generator=None):
        """images: list of uint8 [H, W, 3] numpy arrays / CPU tensors (decoded RGB).  Returns fp32 [B, 3, S, S]."""
        if self.dev.type != "cuda":
            raise L.MMVQAError("DeviceAugment runs on the GPU only (no CPU fallback)")
        S, dev, lib = self.size, self.dev, L.lib()
        B = len(images)
        arrs = [np.ascontiguousarray(im.numpy() if isinstance(im, torch.Tensor) else im, dtype=np.uint8) for im in images]
        for a in arrs:
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("images must be uint8 [H, W, 3]")
        offs = np.cumsum([0] + [a.size for a in arrs])
        host = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs]))
        src = host.to(dev, non_blocking=False)            # the one host->device copy of the batch
        specs = []
        for n, a in enumerate(arrs):
            h, w = a.shape[:2]
            rw, rh = resized_size(w, h, S)
            ox, oy = center_crop_offset(rw, rh, S)
            specs.append((src.data_ptr() + int(offs[n]), h, w, (0, 0, w, h), (rw, rh), (ox, oy)))
        a0 = self._resample(specs)                         # Resize(S) + CenterCrop(S)
        keep = [self._keep, src]
        if self.train:
            if params is None:
                params = sample_params(B, S, self.scale, self.ratio, self.degrees, self.jitter, generator)
            specs = [(a0[n].data_ptr(), S, S, (p["box"][1], p["box"][0], p["box"][3], p["box"][2]), (S, S), (0, 0))
                     for n, p in enumerate(params)]
            a1 = self._resample(specs)                     # RandomResizedCrop
            keep.append(self._keep)
            fix = torch.tensor([rotate_fix(p["angle"], S, S) for p in params], dtype=torch.int32).to(dev)
            a2 = torch.empty_like(a1)
            L.check(lib.mmvqa_aug_rotate(L.stream_ptr(), L.ptr(a1), L.ptr(a2), L.ptr(fix), B, S, S))
            if getattr(self, "debug", False):
                self.stages = dict(resize_crop=a0.clone(), resized_crop=a1.clone(), rotate=a2.clone())
            lsum = torch.zeros(B, dtype=torch.int64, device=dev)
            for rnd in range(4):                           # ColorJitter: image b applies its rnd-th op of the permutation
                ops = [p["order"][rnd] for p in params]
                fac = [(p["brightness"], p["contrast"], p["saturation"], float(hue_shift_u8(p["hue"])))[o]
                       for o, p in zip(ops, params)]
                op_d = torch.tensor(ops, dtype=torch.int32).to(dev)
                fac_d = torch.tensor(fac, dtype=torch.float32).to(dev)
                L.check(lib.mmvqa_aug_jitter_round(L.stream_ptr(), L.ptr(a2), L.ptr(op_d), L.ptr(fac_d), L.ptr(lsum), B, S * S))
                keep += [op_d, fac_d]
            final = a2
            keep += [a0, a1, fix, lsum]
        else:
            final = a0
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
        L.check(lib.mmvqa_aug_to_tensor(L.stream_ptr(), L.ptr(final), L.ptr(out), B, S * S, self.mean, self.std))
        self._keep_all = keep + [final]
        self.last_uint8 = final                             # [B, S, S, 3] after the last byte-valued stage (tests)
        return out

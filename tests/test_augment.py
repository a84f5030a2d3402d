"""Device input pipeline (pretrain/roco_train.py:98-112, vqamed2019/train.py:179-200) against
Pillow itself -- the library whose C code does the reference's pixel arithmetic under torchvision's PIL backend.
Byte-valued stages must be BIT-EXACT; the fp32 ToTensor/Normalize output must be exactly equal as well (same two
IEEE operations).  CPU part: geometry / parameter sampling / the host coefficient routine; GPU part: the kernels."""
import math

import numpy as np
import PIL
import pytest
import torch

from mmvqa_amd import augment as AU
from oracle import augment_oracle as AO


def synth_image(rng, h, w):
    """smooth structure + noise + saturated patches: exercises clipping, hue wrap and anti-aliased down-scaling"""
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 9.0 + yy / 17.0), 127 + 120 * np.cos(xx / 13.0 - yy / 7.0),
                     (xx * 3 + yy * 5) % 256], -1)
    img = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
    img[: h // 8, : w // 8] = 255
    img[-(h // 8):, -(w // 8):] = 0
    img[h // 3: h // 3 + 5] = rng.integers(0, 256, 3)
    return img


def coeffs_np(in_size, in0, in1, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc (bilinear), plain Python doubles"""
    scale = (in1 - in0) / out_size
    fs = scale if scale >= 1.0 else 1.0
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    b = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << 22)) if w[x] < 0 else int(0.5 + w[x] * (1 << 22))
        b[xx] = (xmin, xmax)
    return b, kk, ksize


def test_geometry_matches_the_oracle():
    from PIL import Image
    for (h, w) in [(300, 400), (400, 300), (224, 224), (225, 224), (224, 301), (64, 48), (500, 701), (1023, 767), (333, 999)]:
        rw, rh = AU.resized_size(w, h, 224)
        ox, oy = AU.center_crop_offset(rw, rh, 224)
        assert min(rw, rh) == 224
        ref = AO.resize_center_crop(Image.fromarray(np.zeros((h, w, 3), np.uint8)), 224)
        assert ref.size == (224, 224)
        assert 0 <= ox <= rw - 224 and 0 <= oy <= rh - 224


def test_host_coefficient_routine():
    """mmvqa_resample_coeffs (host C++ in the library) == the restated Pillow routine, down- and up-scaling"""
    for (n_in, n_out) in [(400, 298), (298, 224), (700, 313), (224, 224), (213, 224), (48, 224), (1023, 299), (230, 224)]:
        b, k, ks = AU.coeffs(n_in, 0, n_in, n_out)
        bn, kn, ksn = coeffs_np(n_in, 0, n_in, n_out)
        assert ks == ksn and np.array_equal(b, bn) and np.array_equal(k, kn), (n_in, n_out)
        assert np.all(np.abs(k.sum(1) - (1 << 22)) <= ks)          # normalised to one (up to rounding of each tap)


def test_parameter_sampling():
    g = torch.Generator().manual_seed(5)
    a = AU.sample_params(50, 224, (0.95, 1.05), (0.95, 1.05), 5.0, (0.05,) * 4, g)
    g = torch.Generator().manual_seed(5)
    b = AU.sample_params(50, 224, (0.95, 1.05), (0.95, 1.05), 5.0, (0.05,) * 4, g)
    assert a == b                                                    # same generator state -> same draws
    for p in a:
        i, j, h, w = p["box"]
        assert 0 < h <= 224 and 0 < w <= 224 and 0 <= i <= 224 - h and 0 <= j <= 224 - w
        assert 0.95 * 0.95 - 0.02 <= h * w / 224 ** 2 <= 1.0
        assert -5.0 <= p["angle"] <= 5.0 and sorted(p["order"]) == [0, 1, 2, 3]
        assert 0.95 <= p["brightness"] <= 1.05 and -0.05 <= p["hue"] <= 0.05
    assert len({tuple(p["order"]) for p in a}) > 5
    assert AU.hue_shift_u8(0.05) == 12 and AU.hue_shift_u8(-0.05) == 244 and AU.hue_shift_u8(0.0) == 0
    assert AU.rotate_fix(0.0, 224, 224) == [65536, 0, 32768, 0, 65536, 32768]


SIZES = [(300, 400), (400, 300), (224, 224), (225, 224), (64, 48), (500, 701), (767, 1023), (999, 333), (231, 517)]


@pytest.mark.gpu
def test_val_transform_bit_exact():
    print("Pillow", PIL.__version__)
    rng = np.random.default_rng(0)
    imgs = [synth_image(rng, h, w) for h, w in SIZES]
    aug = AU.DeviceAugment(train=False)
    out = aug(imgs)
    torch.cuda.synchronize()
    got_u8 = aug.last_uint8.cpu().numpy()
    for n, im in enumerate(imgs):
        ref_u8, ref_f = AO.val_transform(im)
        assert np.array_equal(got_u8[n], ref_u8), f"image {n} {im.shape}: {(got_u8[n] != ref_u8).sum()} bytes differ"
        assert torch.equal(out[n].cpu(), ref_f), f"image {n}: float output differs"


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [
    ("roco", dict(scale=(0.95, 1.05), ratio=(0.95, 1.05), degrees=5.0, jitter=(0.05,) * 4)),      # roco_train.py:98-108
    ("vqa", dict(scale=(0.75, 1.25), ratio=(0.75, 1.25), degrees=10.0, jitter=(0.4,) * 4)),       # vqamed2019/train.py:179-190
])
def test_train_transform_bit_exact(name, kw):
    rng = np.random.default_rng(1)
    imgs = [synth_image(rng, h, w) for h, w in SIZES * 3]
    aug = AU.DeviceAugment(train=True, **kw)
    aug.debug = True
    params = AU.sample_params(len(imgs), 224, kw["scale"], kw["ratio"], kw["degrees"], kw["jitter"], torch.Generator().manual_seed(11))
    params[0]["angle"] = 0.0                                          # Image.rotate's copy path
    params[1].update(brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0)
    out = aug(imgs, params=params)
    torch.cuda.synchronize()
    st = {k: v.cpu().numpy() for k, v in aug.stages.items()}
    got_u8 = aug.last_uint8.cpu().numpy()
    from PIL import Image
    for n, (im, p) in enumerate(zip(imgs, params)):
        a0 = AO.resize_center_crop(Image.fromarray(im))
        assert np.array_equal(st["resize_crop"][n], np.array(a0)), (n, "resize+center crop")
        a1 = AO.resized_crop(a0, *p["box"])
        assert np.array_equal(st["resized_crop"][n], np.array(a1)), (n, "random resized crop", p["box"])
        a2 = AO.rotate(a1, p["angle"])
        assert np.array_equal(st["rotate"][n], np.array(a2)), (n, "rotate", p["angle"])
        ref_u8, ref_f = AO.train_transform(im, p)
        bad = int((got_u8[n] != ref_u8).sum())
        assert bad == 0, f"image {n}: {bad} bytes differ after ColorJitter {p}"
        assert torch.equal(out[n].cpu(), ref_f), f"image {n}: float output differs"
    assert out.shape == (len(imgs), 3, 224, 224) and out.dtype == torch.float32 and out.is_cuda


@pytest.mark.gpu
def test_colour_ops_all_values():
    """every byte value through brightness / contrast / saturation with factors on both sides of [0, 1], and every
    hue shift over a colour cube sample: Pillow's truncation, clipping and HSV rounding"""
    from PIL import Image, ImageEnhance
    import ctypes as C
    from mmvqa_amd import _lib as L
    rng = np.random.default_rng(2)
    cube = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing="ij"), -1).reshape(-1, 3)
    img = np.concatenate([cube, rng.integers(0, 256, (224 * 224 - len(cube) % (224 * 224), 3))]).astype(np.uint8)
    npix = (len(img) // 224) * 224
    img = img[:npix].reshape(-1, 224, 3)
    dev = torch.device("cuda")
    lsum = torch.zeros(1, dtype=torch.int64, device=dev)
    for op, facs in ((0, [0.0, 0.37, 0.95, 1.0, 1.05, 1.6, 2.5]), (1, [0.0, 0.6, 1.0, 1.4, 3.0]), (2, [0.0, 0.6, 1.0, 1.4, 2.2]),
                     (3, [-0.5, -0.4, -0.05, 0.0, 0.05, 0.31, 0.5])):
        for f in facs:
            pim = Image.fromarray(img)
            if op == 0:
                ref = ImageEnhance.Brightness(pim).enhance(f)
            elif op == 1:
                ref = ImageEnhance.Contrast(pim).enhance(f)
            elif op == 2:
                ref = ImageEnhance.Color(pim).enhance(f)
            else:
                ref = AO.adjust_hue(pim, f)
            d = torch.from_numpy(img.copy()).to(dev)
            fac = float(AU.hue_shift_u8(f)) if op == 3 else f
            op_d, fac_d = torch.tensor([op], dtype=torch.int32, device=dev), torch.tensor([fac], dtype=torch.float32, device=dev)
            L.check(L.lib().mmvqa_aug_jitter_round(L.stream_ptr(), L.ptr(d), L.ptr(op_d), L.ptr(fac_d), L.ptr(lsum), 1, img.shape[0] * img.shape[1]))
            torch.cuda.synchronize()
            bad = int((d.cpu().numpy() != np.array(ref)).sum())
            assert bad == 0, f"op {op} factor {f}: {bad} bytes differ"

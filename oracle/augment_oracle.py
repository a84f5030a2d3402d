"""Oracle of the image transforms: the reference's torchvision-on-PIL chain
(pretrain/roco_train.py:98-112, vqamed2019/train.py:179-200) executed with Pillow itself.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torchvision is absent from the build image and unpinned by the reference (no requirements file).  Its PIL backend
(torchvision/transforms/_functional_pil.py) is a thin wrapper: every pixel operation below is the Pillow call that
wrapper makes, restated from the published source -- the ARITHMETIC is Pillow's own C code (present in the image,
version printed by tests/test_augment.py), which is what pins the HIP kernels.  The parameter sampling of
RandomResizedCrop / RandomRotation / ColorJitter lives in mmvqa_amd.augment.sample_params (host logic, restated,
"parity unpinned"); this oracle takes the sampled parameters as input.
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image, ImageEnhance


def resize_center_crop(img: Image.Image, size=224):
    """transforms.Resize(size) + transforms.CenterCrop(size)"""
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    if (w, h) != (new_w, new_h):
        img = img.resize((new_w, new_h), Image.BILINEAR)
    top = int(round((new_h - size) / 2.0))
    left = int(round((new_w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def resized_crop(img, i, j, h, w, size=224):
    """F.resized_crop: crop then resize to (size, size), bilinear"""
    img = img.crop((j, i, j + w, i + h))
    return img.resize((size, size), Image.BILINEAR)


def rotate(img, angle):
    """F.rotate(img, angle, NEAREST, expand=False, center=None, fill=0)"""
    return img.rotate(angle, Image.NEAREST, False, None, fillcolor=(0, 0, 0))


def adjust_hue(img, hue_factor):
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore", invalid="ignore"):
        np_h += np.array(np.int64(hue_factor * 255)).astype("uint8")
    h = Image.fromarray(np_h, "L")
    return Image.merge("HSV", (h, s, v)).convert("RGB")


def color_jitter(img, order, brightness, contrast, saturation, hue):
    for fn_id in order:
        if fn_id == 0:
            img = ImageEnhance.Brightness(img).enhance(brightness)
        elif fn_id == 1:
            img = ImageEnhance.Contrast(img).enhance(contrast)
        elif fn_id == 2:
            img = ImageEnhance.Color(img).enhance(saturation)
        else:
            img = adjust_hue(img, hue)
    return img


def to_tensor_normalize(img, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """ToTensor (uint8 HWC -> float CHW / 255) then Normalize ((x - mean) / std), fp32 like torchvision"""
    x = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return (x - m) / s


def train_transform(arr, p, size=224):
    """uint8 [H, W, 3] -> (uint8 [S, S, 3] after the last byte-valued stage, fp32 [3, S, S])"""
    img = resize_center_crop(Image.fromarray(arr), size)
    i, j, h, w = p["box"]
    img = resized_crop(img, i, j, h, w, size)
    img = rotate(img, p["angle"])
    img = color_jitter(img, p["order"], p["brightness"], p["contrast"], p["saturation"], p["hue"])
    return np.array(img), to_tensor_normalize(img)


def val_transform(arr, size=224):
    img = resize_center_crop(Image.fromarray(arr), size)
    return np.array(img), to_tensor_normalize(img)

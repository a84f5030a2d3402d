"""CPU restatement of timm's ``tf_efficientnetv2_m(features_only=True)`` and of the reference's tap head on it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference call sites: models/image_encoding.py:15,26 (``timm.create_model('tf_efficientnetv2_m',
features_only=True, pretrained=True)``, tap widths [24, 48, 80, 176, 512]) and :89-128 (``Timm_EFfNetV2``).
timm is NOT in the build image and the reference has no tests, so the WIRING below is **parity unpinned**: it is
the architecture recalled in SURVEY.md Appendix B (stem 24; stages cn/er/er/ir/ir/ir/ir with repeats 3/5/5/7/14/18/5,
widths 24/48/80/160/176/304/512, strides 1/2/2/2/1/2/1, expansion 1/4/4/4/6/6/6, SE 0.25 of the block input on the
``ir`` stages, SiLU, BatchNorm eps 1e-3, TensorFlow "SAME" padding), cross-checked by the parameter count of the
features-only body (52 200 436, tests/test_abi.py).  The arithmetic is torch's CPU conv2d / batch_norm.
Module / state_dict names follow timm's ``EfficientNetFeatures`` (conv_stem, bn1, blocks.S.B.*).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.mmbert_oracle import serf

BN_EPS = 1e-3
# (type, repeats, stride, expand, out_ch, se)
ARCH = [("cn", 3, 1, 1, 24, 0.0), ("er", 5, 2, 4, 48, 0.0), ("er", 5, 2, 4, 80, 0.0), ("ir", 7, 2, 4, 160, 0.25),
        ("ir", 14, 1, 6, 176, 0.25), ("ir", 18, 2, 6, 304, 0.25), ("ir", 5, 1, 6, 512, 0.25)]
FEATURE_STAGES = (0, 1, 2, 4, 6)


def same_pad(size, k, s):
    """TensorFlow SAME: total padding, the odd element goes to the end"""
    out = math.ceil(size / s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


class Conv2dSame(nn.Conv2d):
    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        pt, pb = same_pad(x.shape[2], k, s)
        pl, pr = same_pad(x.shape[3], k, s)
        return F.conv2d(F.pad(x, (pl, pr, pt, pb)), self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


def bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS)


class ConvBnAct(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = Conv2dSame(cin, cout, 3, stride, bias=False)
        self.bn1 = bn(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        y = F.silu(self.bn1(self.conv(x)))
        return y + x if self.has_skip else y


class EdgeResidual(nn.Module):  # Fused-MBConv
    def __init__(self, cin, cout, stride, exp):
        super().__init__()
        mid = cin * exp
        self.conv_exp = Conv2dSame(cin, mid, 3, stride, bias=False)
        self.bn1 = bn(mid)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn2 = bn(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        y = self.bn2(self.conv_pwl(F.silu(self.bn1(self.conv_exp(x)))))
        return y + x if self.has_skip else y


class SqueezeExcite(nn.Module):
    def __init__(self, ch, rd):
        super().__init__()
        self.conv_reduce = nn.Conv2d(ch, rd, 1, bias=True)
        self.conv_expand = nn.Conv2d(rd, ch, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.conv_expand(F.silu(self.conv_reduce(s))))


class InvertedResidual(nn.Module):  # MBConv
    def __init__(self, cin, cout, stride, exp, se):
        super().__init__()
        mid = cin * exp
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = bn(mid)
        self.conv_dw = Conv2dSame(mid, mid, 3, stride, groups=mid, bias=False)
        self.bn2 = bn(mid)
        self.se = SqueezeExcite(mid, int(round(cin * se)))
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = bn(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_pw(x)))
        y = F.silu(self.bn2(self.conv_dw(y)))
        y = self.bn3(self.conv_pwl(self.se(y)))
        return y + x if self.has_skip else y


class OracleEffNetV2Features(nn.Module):
    def __init__(self, depth_div=1):
        super().__init__()
        self.conv_stem = Conv2dSame(3, 24, 3, 2, bias=False)
        self.bn1 = bn(24)
        stages, cin = [], 24
        for typ, rep, stride, exp, cout, se in ARCH:
            blocks = []
            for b in range(max(1, math.ceil(rep / depth_div))):
                s = stride if b == 0 else 1
                if typ == "cn":
                    blocks.append(ConvBnAct(cin, cout, s))
                elif typ == "er":
                    blocks.append(EdgeResidual(cin, cout, s, exp))
                else:
                    blocks.append(InvertedResidual(cin, cout, s, exp, se))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)

    def forward(self, x):
        x = F.silu(self.bn1(self.conv_stem(x)))
        feats = []
        for i, st in enumerate(self.blocks):
            x = st(x)
            if i in FEATURE_STAGES:
                feats.append(x)
        return feats


class OracleTimmEffNetV2(nn.Module):
    """models/image_encoding.py:43-62,89-115: o = model(img); v_k = GAP(act(conv_k(o[k]))) with
    conv2,3,4,5,7 on o[0..4] (finest map first)."""

    def __init__(self, hidden_size=768, use_relu=False, depth_div=1):
        super().__init__()
        self.model = OracleEffNetV2Features(depth_div)
        self.hidden_size, self.use_relu = hidden_size, use_relu
        for name, c in zip(("conv2", "conv3", "conv4", "conv5", "conv7"), (24, 48, 80, 176, 512)):
            setattr(self, name, nn.Conv2d(c, hidden_size, kernel_size=1, stride=1, bias=False))

    def forward(self, img):
        o = self.model(img)
        act = F.relu if self.use_relu else serf
        return tuple(act(conv(f)).mean(dim=(2, 3)).view(-1, self.hidden_size)
                     for conv, f in zip((self.conv2, self.conv3, self.conv4, self.conv5, self.conv7), o))

"""Helpers for the GPU parity tests: build C-ABI descriptors from torch tensors."""
import ctypes as C

import torch

import mmvqa_amd
from mmvqa_amd import _lib as L


def dev():
    return torch.device("cuda:0")


def P(t):
    return None if t is None else t.data_ptr()


def nhwc(x):
    """NCHW tensor -> contiguous NHWC copy on the GPU"""
    return x.permute(0, 2, 3, 1).contiguous().to(dev())


def from_nhwc(x, N, H, W, Cc):
    return x.view(N, H, W, Cc).permute(0, 3, 1, 2).contiguous().cpu()


def w_ohwi(w):
    """OIHW conv weight -> [O][KH][KW][I] on the GPU"""
    return w.permute(0, 2, 3, 1).contiguous().to(dev())


def linear_geom(d):
    d.g_SH = d.g_SW = d.g_OH = d.g_OW = 1
    d.g_KH = d.g_KW = 1
    d.g_stride = 1
    d.g_pad = 0


def run_igemm(d, kind, nchw=0, tile=0):
    L.check(L.lib().mmvqa_igemm(C.byref(d), kind, nchw, tile, L.stream_ptr()))
    torch.cuda.synchronize()


def conv_desc_fwd(x_nhwc, w_ohwi_t, N, H, W, Cin, Cout, K, stride, pad, out):
    OH = (H + 2 * pad - K) // stride + 1
    OW = (W + 2 * pad - K) // stride + 1
    d = L.GemmDesc()
    d.M, d.N, d.K = N * OH * OW, Cout, K * K * Cin
    d.A, d.a_ld = P(x_nhwc), Cin
    d.B, d.b_ld = P(w_ohwi_t), K * K * Cin
    d.g_SH, d.g_SW, d.g_Cs, d.g_OH, d.g_OW = H, W, Cin, OH, OW
    d.g_KH = d.g_KW = K
    d.g_stride, d.g_pad = stride, pad
    d.C, d.c_ld = P(out), Cout
    return d, OH, OW


def conv_desc_dgrad(dz_nhwc, w_ohwi_t, N, H, W, Cin, Cout, K, stride, pad, out):
    OH = (H + 2 * pad - K) // stride + 1
    OW = (W + 2 * pad - K) // stride + 1
    d = L.GemmDesc()
    d.M, d.N, d.K = N * H * W, Cin, K * K * Cout
    d.A, d.a_ld = P(dz_nhwc), Cout
    d.g_SH, d.g_SW, d.g_Cs, d.g_OH, d.g_OW = OH, OW, Cout, H, W
    d.g_KH = d.g_KW = K
    d.g_stride, d.g_pad = stride, pad
    d.B, d.b_ld, d.b_tapstride = P(w_ohwi_t), K * K * Cin, Cin
    d.C, d.c_ld = P(out), Cin
    return d


def conv_desc_wgrad(dz_nhwc, x_nhwc, N, H, W, Cin, Cout, K, stride, pad, out):
    OH = (H + 2 * pad - K) // stride + 1
    OW = (W + 2 * pad - K) // stride + 1
    d = L.GemmDesc()
    d.M, d.N, d.K = Cout, K * K * Cin, N * OH * OW
    d.A, d.a_ld = P(dz_nhwc), Cout
    d.B, d.b_ld = P(x_nhwc), Cin
    d.g_SH, d.g_SW, d.g_Cs, d.g_OH, d.g_OW = H, W, Cin, OH, OW
    d.g_KH = d.g_KW = K
    d.g_stride, d.g_pad = stride, pad
    d.C, d.c_ld, d.c_atomic = P(out), K * K * Cin, 1
    return d


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, tol, what=""):
    e = relerr(a, b)
    assert e <= tol, f"{what}: rel-to-max error {e:.3e} > {tol:.1e}"

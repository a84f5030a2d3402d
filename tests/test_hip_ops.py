"""GPU parity tests of the individual HIP kernels, called through the C ABI, against torch CPU
fp32 arithmetic of the same op (tolerance 1e-4 relative to the tensor's max, far inside the
1e-3 end-to-end budget; index outputs bit-exact)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hip_helpers import *  # noqa: E402,F401,F403
from oracle import mmbert_oracle as O  # noqa: E402

TOL = 1e-4


def ACT(name):
    return dict(none=L.ACT_NONE, relu=L.ACT_RELU, gelu=L.ACT_GELU, serf=L.ACT_SERF)[name]


def act_cpu(name, x):
    return dict(none=lambda t: t, relu=torch.relu, gelu=O.gelu, serf=O.serf)[name](x)


# ----------------------------------------------------------------------------- linear (igemm FWD/DGRAD/WGRAD)
@pytest.mark.parametrize("M,K,N,act,tile", [(70, 96, 50, "none", 0), (512, 768, 2304, "gelu", 0),
                                           (130, 64, 130, "serf", 1), (33, 100, 257, "relu", 2),
                                           (200, 40, 64, "none", 3), (64, 256, 300, "none", 4),
                                           (300, 1024, 130, "serf", 5), (100, 72, 64, "none", 5)])
def test_linear_fwd(M, K, N, act, tile):
    torch.manual_seed(0)
    x, w, b, r = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N), torch.randn(M, N)
    xd, wd, bd, rd = (t.to(dev()) for t in (x, w, b, r))
    ldy = (N + 3) & ~3
    y = torch.zeros(M, ldy, device=dev())
    pre = torch.zeros(M, ldy, device=dev())
    d = L.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A, d.a_ld, d.g_Cs = P(xd), K, K
    d.B, d.b_ld = P(wd), K
    linear_geom(d)
    d.C, d.c_ld, d.Cpre, d.bias, d.act = P(y), ldy, P(pre), P(bd), ACT(act)
    d.R, d.r_ld = P(rd), N
    run_igemm(d, L.KIND_FWD, 0, tile)
    ref_pre = x @ w.T + b
    assert_close(pre[:, :N], ref_pre, TOL, "pre")
    assert_close(y[:, :N], act_cpu(act, ref_pre) + r, TOL, "y")


def test_linear_bwd():
    torch.manual_seed(1)
    M, K, N = 96, 72, 130   # N not a multiple of 4: padded leading dimension (the vocab case)
    x = torch.randn(M, K, requires_grad=True)
    w = (torch.randn(N, K) / math.sqrt(K)).requires_grad_(True)
    pre = x @ w.T
    gy = torch.randn(M, N)
    O.serf(pre).backward(gy)
    p2 = pre.detach().clone().requires_grad_(True)
    O.serf(p2).backward(gy)
    gpre = p2.grad                      # gradient wrt pre = gy * serf'(pre)
    ld = (N + 3) & ~3
    gp = torch.full((M, ld), float("nan"))   # pad columns hold garbage on purpose
    gp[:, :N] = gpre
    gpd, xd, wd = gp.to(dev()), x.detach().to(dev()), w.detach().to(dev())
    dx = torch.zeros(M, K, device=dev())
    d = L.GemmDesc()
    d.M, d.N, d.K = M, K, N
    d.A, d.a_ld, d.g_Cs = P(gpd), ld, N
    d.B, d.b_ld = P(wd), K
    linear_geom(d)
    d.C, d.c_ld = P(dx), K
    run_igemm(d, L.KIND_DGRAD)
    assert_close(dx, x.grad, TOL, "dx")
    dw = torch.zeros(N, K, device=dev())
    d = L.GemmDesc()
    d.M, d.N, d.K = N, K, M
    d.A, d.a_ld = P(gpd), ld
    d.B, d.b_ld, d.g_Cs = P(xd), K, K
    linear_geom(d)
    d.C, d.c_ld, d.c_atomic = P(dw), K, 1
    run_igemm(d, L.KIND_WGRAD)
    assert_close(dw, w.grad, TOL, "dw")


def test_dgrad_fused_act_backward():
    """fc2-dgrad epilogue: dpre = (dy W2) * act'(pre), with the bias-gradient column sums"""
    torch.manual_seed(14)
    M, Hh, F4 = 64, 48, 132
    dy, w2, pre = torch.randn(M, Hh), torch.randn(Hh, F4) / 7, torch.randn(M, F4) * 2
    for act in ("gelu", "serf"):
        pr = pre.clone().requires_grad_(True)
        (act_cpu(act, pr) @ w2.T).backward(dy)
        out = torch.zeros(M, F4, device=dev())
        cs = torch.zeros(F4, device=dev())
        dyd, w2d, pred = dy.to(dev()), w2.to(dev()), pre.to(dev())
        d = L.GemmDesc()
        d.M, d.N, d.K = M, F4, Hh
        d.A, d.a_ld, d.g_Cs = P(dyd), Hh, Hh
        d.B, d.b_ld = P(w2d), F4
        linear_geom(d)
        d.C, d.c_ld = P(out), F4
        d.dact, d.Pre, d.pre_ld, d.colsum = ACT(act), P(pred), F4, P(cs)
        run_igemm(d, L.KIND_DGRAD)
        assert_close(out, pr.grad, TOL, "dact " + act)
        assert_close(cs, pr.grad.sum(0), TOL, "colsum " + act)


def test_act_golden_edge_replay(golden_dir):
    """tests/golden/act.npz -- x incl. {-100, -50, -20, 0, 20, 49.9, 50, 50.1, 80, 1e4} with the REFERENCE's SERF
    (models/serf.py:23-24) and GELU (models/transformer.py:7-8) values and derivatives -- through the HIP GEMM
    epilogues: an identity GEMM (exact in fp32 MFMA arithmetic) feeds x to act() / act'()."""
    import numpy as np
    g = dict(np.load(f"{golden_dir}/act.npz"))
    x = torch.from_numpy(g["x"]).float()
    M, Kd = x.numel(), 4
    X = torch.zeros(M, Kd)
    X[:, 0] = x
    Xd, Id = X.to(dev()), torch.eye(Kd).to(dev())
    e0 = torch.zeros(M, Kd)
    e0[:, 0] = 1.0
    e0d = e0.to(dev())
    for act in ("serf", "gelu"):
        y, pre = torch.zeros(M, Kd, device=dev()), torch.zeros(M, Kd, device=dev())
        d = L.GemmDesc()
        d.M, d.N, d.K = M, Kd, Kd
        d.A, d.a_ld, d.g_Cs = P(Xd), Kd, Kd
        d.B, d.b_ld = P(Id), Kd
        linear_geom(d)
        d.C, d.c_ld, d.Cpre, d.act = P(y), Kd, P(pre), ACT(act)
        run_igemm(d, L.KIND_FWD)
        assert torch.equal(pre[:, 0].cpu(), x), "identity GEMM must reproduce x bit for bit"
        ref = torch.from_numpy(g[act]).double()
        err = ((y[:, 0].cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        assert err <= 2e-6, f"{act} forward vs reference: {err:.2e}"
        out = torch.zeros(M, Kd, device=dev())
        d = L.GemmDesc()
        d.M, d.N, d.K = M, Kd, Kd
        d.A, d.a_ld, d.g_Cs = P(e0d), Kd, Kd
        d.B, d.b_ld = P(Id), Kd
        linear_geom(d)
        d.C, d.c_ld = P(out), Kd
        d.dact, d.Pre, d.pre_ld = ACT(act), P(Xd), Kd
        run_igemm(d, L.KIND_DGRAD)
        dref = torch.from_numpy(g["d" + act]).double()
        err = ((out[:, 0].cpu().double() - dref).abs() / dref.abs().clamp_min(1.0)).max().item()
        assert err <= 3e-6, f"{act} derivative vs reference: {err:.2e}"
        edge = x.abs() >= 20       # the ten hand-picked edge values: clamp at 50, saturation, underflow
        assert int(edge.sum()) >= 9 and torch.isfinite(y[:, 0]).all() and torch.isfinite(out[:, 0]).all()


# ----------------------------------------------------------------------------- convolution
CONVS = [  # N, H, W, Cin, Cout, K, stride, pad
    (2, 9, 9, 8, 16, 3, 1, 1), (2, 9, 9, 8, 16, 3, 2, 1), (3, 8, 8, 16, 24, 1, 1, 0), (2, 9, 9, 16, 8, 1, 2, 0),
    (2, 14, 14, 64, 64, 3, 1, 1), (1, 7, 7, 40, 72, 3, 2, 1),
    # channel counts that are multiples of the K-tile: the uniform-tap loaders (1x1, strided 3x3, strided 1x1)
    (2, 10, 10, 128, 64, 1, 1, 0), (2, 12, 12, 64, 128, 3, 2, 1), (2, 12, 12, 64, 128, 1, 2, 0), (3, 8, 8, 128, 128, 3, 1, 1),
]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_bwd(cfg):
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(2)
    x_raw = torch.randn(N, Cin, H, W)
    sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    w = torch.randn(Cout, Cin, K, K) / math.sqrt(Cin * K * K)
    # forward: z = conv(relu(x*sc+sh)) with fused statistics
    a = torch.relu(x_raw * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    z_ref = F.conv2d(a, wr, stride=s, padding=p)
    OH, OW = z_ref.shape[2:]
    xd, wd = nhwc(x_raw), w_ohwi(w)
    scd, shd = sc.to(dev()), sh.to(dev())
    z = torch.zeros(N * OH * OW, Cout, device=dev())
    stat = torch.zeros(L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=dev())
    d, _, _ = conv_desc_fwd(xd, wd, N, H, W, Cin, Cout, K, s, p, z)
    d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
    d.stat1, d.stat_bwd = P(stat), 0
    run_igemm(d, L.KIND_FWD)
    assert_close(from_nhwc(z, N, OH, OW, Cout), z_ref, TOL, "z")
    st = stat.sum(0).cpu()
    assert_close(st[:, 0], z_ref.sum(dim=(0, 2, 3)).double(), 1e-5, "sum")
    assert_close(st[:, 1], (z_ref.double() ** 2).sum(dim=(0, 2, 3)), 1e-5, "sumsq")
    # backward through dz = Pc*G + Qc*z + Rc (BatchNorm-backward prologue with arbitrary coefficients)
    G = torch.randn_like(z_ref)
    Pc, Qc, Rc = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1
    dz = G * Pc[None, :, None, None] + z_ref.detach() * Qc[None, :, None, None] + Rc[None, :, None, None]
    z_ref.backward(dz)
    Gd, zd = nhwc(G), z
    coef = [t.to(dev()) for t in (Pc, Qc, Rc)]
    # dgrad with ReLU mask of the producer + BN-backward statistics on x_raw
    mu, istd = torch.randn(Cin) * 0.1, torch.rand(Cin) + 0.5
    dx = torch.zeros(N * H * W, Cin, device=dev())
    bst = torch.zeros(L.STAT_SLOTS, Cin, 2, dtype=torch.float64, device=dev())
    d = conv_desc_dgrad(Gd, wd, N, H, W, Cin, Cout, K, s, p, dx)
    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(zd), L.PRO_DZ, P(coef[0]), P(coef[1]), P(coef[2])
    d.Mk, d.mk_ld, d.mk_s, d.mk_b = P(xd), Cin, P(scd), P(shd)
    mud, isd = mu.to(dev()), istd.to(dev())
    d.stat1, d.stat_bwd, d.Z1, d.z1_ld, d.mean1, d.invstd1 = P(bst), 1, P(xd), Cin, P(mud), P(isd)
    run_igemm(d, L.KIND_DGRAD)
    mask = (x_raw * sc[None, :, None, None] + sh[None, :, None, None] > 0).float()
    g_ref = a.grad * mask
    assert_close(from_nhwc(dx, N, H, W, Cin), g_ref, TOL, "dgrad")
    xhat = (x_raw - mu[None, :, None, None]) * istd[None, :, None, None]
    bs = bst.sum(0).cpu()
    assert_close(bs[:, 0], g_ref.sum(dim=(0, 2, 3)).double(), 1e-4, "sum g")
    assert_close(bs[:, 1], (g_ref * xhat).sum(dim=(0, 2, 3)).double(), 1e-4, "sum g xhat")
    # wgrad with the producer's BN+ReLU applied to x on the fly
    dw = torch.zeros(Cout, K * K * Cin, device=dev())
    d = conv_desc_wgrad(Gd, xd, N, H, W, Cin, Cout, K, s, p, dw)
    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(zd), L.PRO_DZ, P(coef[0]), P(coef[1]), P(coef[2])
    d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
    run_igemm(d, L.KIND_WGRAD)
    assert_close(dw.view(Cout, K, K, Cin).permute(0, 3, 1, 2), wr.grad, TOL, "wgrad")
    if K > 1 and s == 1 and OH == H and Cin % 64 == 0:
        # the same weight gradient through the uniform-tap loaders: per-pixel tap-validity table + 64x64 tiles
        tab = torch.zeros(N * OH * OW, dtype=torch.int32, device=dev())
        L.check(L.lib().mmvqa_pixmask(L.stream_ptr(), P(tab), N, OH, OW, H, W, K, K, s, p))
        ref_tab = torch.zeros(N, OH, OW, dtype=torch.int32)
        for kh in range(K):
            for kw in range(K):
                ys, xs = torch.arange(OH) * s - p + kh, torch.arange(OW) * s - p + kw
                ok = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
                ref_tab += (ok.int() << (kh * K + kw))[None]
        assert torch.equal(tab.cpu().view(N, OH, OW), ref_tab)
        d.pixmask = P(tab)
        for tile in (3, 5):
            dw.zero_()
            run_igemm(d, L.KIND_WGRAD, tile=tile)
            assert_close(dw.view(Cout, K, K, Cin).permute(0, 3, 1, 2), wr.grad, TOL, f"wgrad (pixel table, tile {tile})")


@pytest.mark.parametrize("cfg,tile", [((2, 7, 7, 256, 72, 1, 1, 0), 3), ((2, 7, 7, 256, 72, 1, 1, 0), 5), ((1, 7, 7, 128, 40, 3, 1, 1), 6),
                                      ((3, 6, 6, 192, 52, 1, 1, 0), 3)])
@pytest.mark.parametrize("splitk", [2, 3, 8])
@pytest.mark.parametrize("ticket", [False, True])
def test_conv_split_k_with_finishing_launch(cfg, tile, splitk, ticket):
    """forward / data-gradient products with few output tiles and a long contraction: K split over workgroups into a
    caller-provided scratch (sk_ws) + the finishing launch that sums the partial tiles and runs the WHOLE epilogue
    (BN statistics forward and backward, ReLU mask, residual).  ticket=True: with the caller's tickets (sk_cnt) there is
    no second launch -- the last workgroup of a tile to arrive sums the partial tiles and runs the epilogue; the
    tickets must be zero again afterwards.  Same answers as the single launch and as torch; ragged
    N (72, 52, 40: not multiples of 64), a split count that does not divide the K-tiles, and a
    scratch too small for the requested split (the launcher lowers the split instead of overrunning it)."""
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(5)
    x_raw = torch.randn(N, Cin, H, W)
    sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    w = torch.randn(Cout, Cin, K, K) / math.sqrt(Cin * K * K)
    a = torch.relu(x_raw * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    z_ref = F.conv2d(a, w, stride=s, padding=p)
    OH, OW = z_ref.shape[2:]
    xd, wd = nhwc(x_raw), w_ohwi(w)
    scd, shd = sc.to(dev()), sh.to(dev())
    M = N * OH * OW
    pad = lambda m, n: ((m + 63) // 64) * ((n + 63) // 64) * 4096   # the ticketed form keeps whole tiles
    ws = torch.full((splitk * (pad(M, Cout) if ticket else M * Cout) + 16,), float("nan"), device=dev())   # NaN: every partial element must be written
    cnt = torch.zeros(256, dtype=torch.int32, device=dev())

    def tickets(d):
        if ticket:
            d.sk_cnt, d.sk_cnt_n = P(cnt), cnt.numel()
    res = {}
    for sk in (1, splitk):
        z = torch.zeros(M, Cout, device=dev())
        stat = torch.zeros(L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=dev())
        d, _, _ = conv_desc_fwd(xd, wd, N, H, W, Cin, Cout, K, s, p, z)
        d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
        d.stat1, d.stat_bwd = P(stat), 0
        d.splitk, d.sk_ws, d.sk_ws_floats = sk, P(ws), ws.numel() - 16
        tickets(d)
        run_igemm(d, L.KIND_FWD, tile=tile)
        res[sk] = (z, stat.sum(0).cpu())
    assert_close(from_nhwc(res[splitk][0], N, OH, OW, Cout), z_ref, TOL, "z (split-K)")
    assert_close(res[splitk][0], res[1][0], 1e-5, "z split vs single launch")
    assert_close(res[splitk][1][:, 0], z_ref.sum(dim=(0, 2, 3)).double(), 1e-5, "sum")
    assert_close(res[splitk][1][:, 1], (z_ref.double() ** 2).sum(dim=(0, 2, 3)), 1e-5, "sumsq")
    # data gradient of the transposed role: gradient wrt a, ReLU mask of the producer, residual, backward statistics
    G = torch.randn_like(z_ref)
    z_ref.backward(G)
    Gd = nhwc(G)
    Rr = torch.randn(N * H * W, Cin)
    Rd = Rr.to(dev())
    mu, istd = torch.randn(Cin) * 0.1, torch.rand(Cin) + 0.5
    mud, isd = mu.to(dev()), istd.to(dev())
    M2 = N * H * W
    full = splitk * (pad(M2, Cin) if ticket else M2 * Cin)
    ws2 = torch.full((full,), float("nan"), device=dev())
    out = {}
    for sk, cap in ((1, full), (splitk, full), (splitk, 2 * M2 * Cin)):
        dx = torch.zeros(M2, Cin, device=dev())
        bst = torch.zeros(L.STAT_SLOTS, Cin, 2, dtype=torch.float64, device=dev())
        d = conv_desc_dgrad(Gd, wd, N, H, W, Cin, Cout, K, s, p, dx)
        d.Mk, d.mk_ld, d.mk_s, d.mk_b = P(xd), Cin, P(scd), P(shd)
        d.R, d.r_ld = P(Rd), Cin
        d.stat1, d.stat_bwd, d.Z1, d.z1_ld, d.mean1, d.invstd1 = P(bst), 1, P(xd), Cin, P(mud), P(isd)
        d.splitk, d.sk_ws, d.sk_ws_floats = sk, P(ws2), cap
        tickets(d)
        run_igemm(d, L.KIND_DGRAD, tile=tile)
        out[(sk, cap)] = (dx, bst.sum(0).cpu())
    assert int(cnt.abs().sum()) == 0, "tickets not back at zero"
    mask = (x_raw * sc[None, :, None, None] + sh[None, :, None, None] > 0).float()
    g_ref = (a.grad + Rr.view(N, H, W, Cin).permute(0, 3, 1, 2)) * mask
    xhat = (x_raw - mu[None, :, None, None]) * istd[None, :, None, None]
    for key, (dx, bs) in out.items():
        assert_close(from_nhwc(dx, N, H, W, Cin), g_ref, TOL, f"dgrad {key}")
        assert_close(bs[:, 0], g_ref.sum(dim=(0, 2, 3)).double(), 1e-4, f"sum g {key}")
        assert_close(bs[:, 1], (g_ref * xhat).sum(dim=(0, 2, 3)).double(), 1e-4, f"sum g xhat {key}")
    # without a scratch a split forward product is refused, not silently run unsplit
    d, _, _ = conv_desc_fwd(xd, wd, N, H, W, Cin, Cout, K, s, p, res[1][0])
    d.splitk = 2
    assert L.lib().mmvqa_igemm(C.byref(d), L.KIND_FWD, 0, tile, L.stream_ptr()) != 0


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("cfg", [(2, 14, 14, 128, 192, 3, 1, 1), (3, 9, 9, 64, 80, 1, 1, 0), (2, 12, 12, 40, 72, 3, 2, 1)])
def test_conv_every_tile_variant(cfg, tile):
    """every tile variant the tuner can pick (128x128, 128x64, 64x64 with 64-deep K-tiles, 64x128, the 8-wave form, 64x64
    with 32-deep K-tiles) on the uniform-tap loaders (channel counts that are multiples of the K-tile) and on the general
    ones (40 -> 72 channels, stride 2): forward with BN+ReLU prologue and statistics, data gradient with mask and backward
    statistics, weight gradient -- each against torch"""
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(11)
    x_raw = torch.randn(N, Cin, H, W)
    sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    w = torch.randn(Cout, Cin, K, K) / math.sqrt(Cin * K * K)
    a = torch.relu(x_raw * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    z_ref = F.conv2d(a, wr, stride=s, padding=p)
    OH, OW = z_ref.shape[2:]
    xd, wd = nhwc(x_raw), w_ohwi(w)
    scd, shd = sc.to(dev()), sh.to(dev())
    z = torch.zeros(N * OH * OW, Cout, device=dev())
    stat = torch.zeros(L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=dev())
    d, _, _ = conv_desc_fwd(xd, wd, N, H, W, Cin, Cout, K, s, p, z)
    d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
    d.stat1, d.stat_bwd = P(stat), 0
    run_igemm(d, L.KIND_FWD, tile=tile)
    assert_close(from_nhwc(z, N, OH, OW, Cout), z_ref, TOL, "z")
    st = stat.sum(0).cpu()
    assert_close(st[:, 0], z_ref.sum(dim=(0, 2, 3)).double(), 1e-5, "sum")
    assert_close(st[:, 1], (z_ref.double() ** 2).sum(dim=(0, 2, 3)), 1e-5, "sumsq")
    G = torch.randn_like(z_ref)
    z_ref.backward(G)
    Gd = nhwc(G)
    mu, istd = torch.randn(Cin) * 0.1, torch.rand(Cin) + 0.5
    mud, isd = mu.to(dev()), istd.to(dev())
    dx = torch.zeros(N * H * W, Cin, device=dev())
    bst = torch.zeros(L.STAT_SLOTS, Cin, 2, dtype=torch.float64, device=dev())
    d = conv_desc_dgrad(Gd, wd, N, H, W, Cin, Cout, K, s, p, dx)
    d.Mk, d.mk_ld, d.mk_s, d.mk_b = P(xd), Cin, P(scd), P(shd)
    d.stat1, d.stat_bwd, d.Z1, d.z1_ld, d.mean1, d.invstd1 = P(bst), 1, P(xd), Cin, P(mud), P(isd)
    run_igemm(d, L.KIND_DGRAD, tile=tile)
    mask = (x_raw * sc[None, :, None, None] + sh[None, :, None, None] > 0).float()
    g_ref = a.grad * mask
    assert_close(from_nhwc(dx, N, H, W, Cin), g_ref, TOL, "dgrad")
    xhat = (x_raw - mu[None, :, None, None]) * istd[None, :, None, None]
    bs = bst.sum(0).cpu()
    assert_close(bs[:, 0], g_ref.sum(dim=(0, 2, 3)).double(), 1e-4, "sum g")
    assert_close(bs[:, 1], (g_ref * xhat).sum(dim=(0, 2, 3)).double(), 1e-4, "sum g xhat")
    dw = torch.zeros(Cout, K * K * Cin, device=dev())
    d = conv_desc_wgrad(Gd, xd, N, H, W, Cin, Cout, K, s, p, dw)
    d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
    run_igemm(d, L.KIND_WGRAD, tile=tile)
    assert_close(dw.view(Cout, K, K, Cin).permute(0, 3, 1, 2), wr.grad, TOL, "wgrad")


def test_stem_conv():
    torch.manual_seed(3)
    N, H, W, Cout = 2, 20, 22, 16
    img = torch.randn(N, 3, H, W)
    w = (torch.randn(Cout, 3, 7, 7) / 12).requires_grad_(True)
    z_ref = F.conv2d(img, w, stride=2, padding=3)
    OH, OW = z_ref.shape[2:]
    imgd, wd = img.to(dev()), w_ohwi(w.detach())
    z = torch.zeros(N * OH * OW, Cout, device=dev())
    d = L.GemmDesc()
    d.M, d.N, d.K = N * OH * OW, Cout, 147
    d.A, d.g_nchw = P(imgd), 1
    d.g_SH, d.g_SW, d.g_Cs, d.g_OH, d.g_OW = H, W, 3, OH, OW
    d.g_KH = d.g_KW = 7
    d.g_stride, d.g_pad = 2, 3
    d.B, d.b_ld, d.C, d.c_ld = P(wd), 147, P(z), Cout
    run_igemm(d, L.KIND_FWD, 1)
    assert_close(from_nhwc(z, N, OH, OW, Cout), z_ref, TOL, "stem z")
    G = torch.randn_like(z_ref)
    z_ref.backward(G)
    Gd = nhwc(G)
    one, zero = torch.ones(Cout, device=dev()), torch.zeros(Cout, device=dev())
    dw = torch.zeros(Cout, 147, device=dev())
    d = L.GemmDesc()
    d.M, d.N, d.K = Cout, 147, N * OH * OW
    d.A, d.A2, d.a_ld, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(Gd), P(z), Cout, L.PRO_DZ, P(one), P(zero), P(zero)
    d.B, d.g_nchw = P(imgd), 1
    d.g_SH, d.g_SW, d.g_Cs, d.g_OH, d.g_OW = H, W, 3, OH, OW
    d.g_KH = d.g_KW = 7
    d.g_stride, d.g_pad = 2, 3
    d.C, d.c_ld, d.c_atomic = P(dw), 147, 1
    run_igemm(d, L.KIND_WGRAD, 1)
    assert_close(dw.view(Cout, 7, 7, 3).permute(0, 3, 1, 2), w.grad, TOL, "stem dw")


def test_tap_fwd_bwd():
    """models/image_encoding.py:53-62: v = mean_hw(act(conv1x1(fmap)))"""
    torch.manual_seed(4)
    N, HW, Cc, Hd = 3, 49, 40, 96
    f = torch.randn(N, HW, Cc, requires_grad=True)
    w = (torch.randn(Hd, Cc) / math.sqrt(Cc)).requires_grad_(True)
    v_ref = O.serf(f @ w.T).mean(1)
    dv = torch.randn(N, Hd)
    v_ref.backward(dv)
    fd, wd, dvd = f.detach().reshape(N * HW, Cc).to(dev()), w.detach().to(dev()), dv.to(dev())
    v = torch.zeros(N, Hd, device=dev())
    d = L.GemmDesc()
    d.M, d.N, d.K = N * HW, Hd, Cc
    d.A, d.a_ld, d.g_Cs, d.B, d.b_ld = P(fd), Cc, Cc, P(wd), Cc
    linear_geom(d)
    d.epi_mode, d.act, d.tap_HW, d.tap_out, d.C, d.c_ld = L.EPI_TAP_FWD, L.ACT_SERF, HW, P(v), P(v), Hd
    run_igemm(d, L.KIND_FWD)
    assert_close(v, v_ref, TOL, "tap v")
    du = torch.zeros(N * HW, Hd, device=dev())
    d.epi_mode, d.tap_out, d.tap_dv, d.C = L.EPI_TAP_BWD, None, P(dvd), P(du)
    run_igemm(d, L.KIND_FWD)
    pre = (f.detach() @ w.detach().T).requires_grad_(True)
    O.serf(pre).mean(1).backward(dv)
    assert_close(du.view(N, HW, Hd), pre.grad, TOL, "tap du")


# ----------------------------------------------------------------------------- BatchNorm pieces
def test_bn_coef_and_add_relu():
    torch.manual_seed(5)
    N, Cc, H, W = 4, 16, 6, 6
    z = torch.randn(N, Cc, H, W) * 2 + 1
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    bn.train()
    y_ref = bn(z)
    M = N * H * W
    stat = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64)
    stat[3, :, 0] = z.double().sum(dim=(0, 2, 3))
    stat[5, :, 1] = (z.double() ** 2).sum(dim=(0, 2, 3))
    statd = stat.to(dev())
    rm, rv = torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())
    nbt = torch.zeros(1, dtype=torch.int64, device=dev())
    out = [torch.zeros(Cc, device=dev()) for _ in range(4)]
    g, b = bn.weight.detach().to(dev()), bn.bias.detach().to(dev())
    L.check(L.lib().mmvqa_bn_coef_fwd(L.stream_ptr(), P(statd), Cc, float(M), 1e-5, P(g), P(b), P(rm), P(rv), P(nbt),
                                      0.1, 1, 1, *[P(t) for t in out]))
    torch.cuda.synchronize()
    assert_close(rm, bn.running_mean, 1e-5, "running_mean")
    assert_close(rv, bn.running_var, 1e-5, "running_var")
    assert int(nbt) == 1
    zd = nhwc(z)
    idn = torch.randn(N, Cc, H, W)
    idnd = nhwc(idn)
    o = torch.zeros(M, Cc, device=dev())
    L.check(L.lib().mmvqa_bn_add_relu(L.stream_ptr(), P(zd), P(out[0]), P(out[1]), P(idnd), None, None, P(o), M, Cc))
    torch.cuda.synchronize()
    assert_close(from_nhwc(o, N, H, W, Cc), torch.relu(y_ref + idn), TOL, "bn_add_relu")
    # k-fold update rule (quirk 7)
    rm2, rv2 = torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())
    L.check(L.lib().mmvqa_bn_coef_fwd(L.stream_ptr(), P(statd), Cc, float(M), 1e-5, P(g), P(b), P(rm2), P(rv2), P(nbt),
                                      0.1, 5, 1, *[P(t) for t in out]))
    bn2 = torch.nn.BatchNorm2d(Cc).train()
    for _ in range(5):
        bn2(z)
    assert_close(rm2, bn2.running_mean, 1e-5, "running_mean x5")
    assert_close(rv2, bn2.running_var, 1e-5, "running_var x5")
    assert int(nbt) == 6


PERSIST = [  # (N, H, W, Cin, Cout, K, stride, pad), tile, workgroups
    ((2, 14, 14, 128, 192, 3, 1, 1), 3, 16), ((2, 14, 14, 128, 192, 3, 1, 1), 5, 29), ((2, 14, 14, 128, 192, 3, 1, 1), 6, 64),
    ((3, 9, 9, 256, 72, 1, 1, 0), 5, 11), ((2, 12, 12, 64, 128, 3, 2, 1), 6, 37), ((2, 10, 10, 128, 64, 1, 1, 0), 3, 8),
    ((1, 7, 7, 192, 200, 3, 1, 1), 5, 256),
]


@pytest.mark.parametrize("cfg,tile,G", PERSIST)
def test_conv_persistent_form(cfg, tile, G):
    """mmvqa_gemm_desc.persist: G workgroups walk equal shares of the launch's K-tile iterations; tiles whose K range is
    cut over several workgroups are completed by the last arriver (tickets + partial tiles in sk_ws), accumulating
    products just add.  Grids that do not divide (G = 29, 37, 11 ...), more workgroups than tiles, the fused BatchNorm
    prologues / statistics epilogues and the folded coefficients all go through it; results against torch and against
    the one-workgroup-per-tile launch; the tickets must read zero afterwards (the next launch relies on it)."""
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(21)
    x_raw = torch.randn(N, Cin, H, W)
    sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    w = torch.randn(Cout, Cin, K, K) / math.sqrt(Cin * K * K)
    a = torch.relu(x_raw * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    z_ref = F.conv2d(a, wr, stride=s, padding=p)
    OH, OW = z_ref.shape[2:]
    xd, wd, scd, shd = nhwc(x_raw), w_ohwi(w), sc.to(dev()), sh.to(dev())
    ws = torch.zeros(8 << 20, device=dev())
    cnt = torch.zeros(4096, dtype=torch.int32, device=dev())

    def fwd(persist):
        z = torch.zeros(N * OH * OW, Cout, device=dev())
        stat = torch.zeros(L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=dev())
        d, _, _ = conv_desc_fwd(xd, wd, N, H, W, Cin, Cout, K, s, p, z)
        d.a_pro, d.a_c0, d.a_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
        d.stat1, d.stat_bwd, d.stat_slots = P(stat), 0, 4
        d.sk_ws, d.sk_ws_floats, d.sk_cnt, d.sk_cnt_n, d.persist, d.splitk = P(ws), ws.numel(), P(cnt), cnt.numel(), persist, 1
        run_igemm(d, L.KIND_FWD, tile=tile)
        assert float(stat[4:].abs().sum()) == 0.0 and float(stat[:4].abs().sum()) > 0.0   # stat_slots = 4: replicas 0..3 only
        return z, stat.sum(0).cpu()

    z0, st0 = fwd(0)
    z1, st1 = fwd(G)
    assert int(cnt.abs().sum()) == 0, "tickets not back to zero"
    assert_close(from_nhwc(z1, N, OH, OW, Cout), z_ref, TOL, "persistent forward vs torch")
    assert_close(z1, z0, 2e-6, "persistent vs one workgroup per tile")
    assert_close(st1[:, 0], st0[:, 0], 1e-6, "sum")
    assert_close(st1[:, 1], (z_ref.double() ** 2).sum(dim=(0, 2, 3)), 1e-5, "sumsq")
    # data gradient with the BatchNorm-backward prologue, ReLU mask and backward statistics
    Gr = torch.randn_like(z_ref)
    Pc, Qc, Rc = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1
    dz = Gr * Pc[None, :, None, None] + z_ref.detach() * Qc[None, :, None, None] + Rc[None, :, None, None]
    z_ref.backward(dz)
    Gd, coef = nhwc(Gr), [t.to(dev()) for t in (Pc, Qc, Rc)]
    mu, istd = (torch.randn(Cin) * 0.1).to(dev()), (torch.rand(Cin) + 0.5).to(dev())

    def dgrad(persist):
        dx = torch.zeros(N * H * W, Cin, device=dev())
        bst = torch.zeros(L.STAT_SLOTS, Cin, 2, dtype=torch.float64, device=dev())
        d = conv_desc_dgrad(Gd, wd, N, H, W, Cin, Cout, K, s, p, dx)
        d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z0), L.PRO_DZ, P(coef[0]), P(coef[1]), P(coef[2])
        d.Mk, d.mk_ld, d.mk_s, d.mk_b = P(xd), Cin, P(scd), P(shd)
        d.stat1, d.stat_bwd, d.Z1, d.z1_ld, d.mean1, d.invstd1 = P(bst), 1, P(xd), Cin, P(mu), P(istd)
        d.sk_ws, d.sk_ws_floats, d.sk_cnt, d.sk_cnt_n, d.persist, d.splitk = P(ws), ws.numel(), P(cnt), cnt.numel(), persist, 1
        run_igemm(d, L.KIND_DGRAD, tile=tile)
        return dx, bst.sum(0).cpu()

    dx0, b0 = dgrad(0)
    dx1, b1 = dgrad(G)
    assert int(cnt.abs().sum()) == 0
    mask = (x_raw * sc[None, :, None, None] + sh[None, :, None, None] > 0).float()
    assert_close(from_nhwc(dx1, N, H, W, Cin), a.grad * mask, TOL, "persistent dgrad vs torch")
    assert_close(dx1, dx0, 2e-6, "persistent dgrad vs one workgroup per tile")
    assert_close(b1, b0, 1e-6, "backward statistics")

    def wgrad(persist):
        dw = torch.zeros(Cout, K * K * Cin, device=dev())
        d = conv_desc_wgrad(Gd, xd, N, H, W, Cin, Cout, K, s, p, dw)
        d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z0), L.PRO_DZ, P(coef[0]), P(coef[1]), P(coef[2])
        d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(scd), P(shd)
        d.persist, d.splitk = persist, 1
        run_igemm(d, L.KIND_WGRAD, tile=tile)
        return dw

    dw1 = wgrad(G)
    assert_close(dw1.view(Cout, K, K, Cin).permute(0, 3, 1, 2), wr.grad, TOL, "persistent wgrad vs torch")
    assert_close(dw1, wgrad(0), 1e-5, "persistent wgrad vs split grid")


def _spread(sums, slots):
    """per-channel (sum, sum2) pairs scattered over the first `slots` statistic replicas, as the producers' atomics leave them"""
    Cc = sums.shape[0]
    stat = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64)
    w = torch.rand(slots, Cc, 2, dtype=torch.float64)
    w = w / w.sum(0, keepdim=True)
    stat[:slots] = w * sums[None]
    return stat.to(dev())


def _fold(stat, slots, bwd, publish, count, gamma, **kw):
    f = L.BnFold()
    f.stat, f.slots, f.bwd, f.publish, f.count, f.gamma = P(stat), slots, bwd, publish, float(count), P(gamma)
    f.eps, f.reps, f.keep = kw.get("eps", 1e-5), kw.get("reps", 1), (1.0 - 0.1) ** kw.get("reps", 1)
    for k in ("beta", "mean", "invstd", "out0", "out1", "out2", "out3", "run_mean", "run_var", "nbt", "dgamma", "dbeta"):
        if k in kw:
            setattr(f, k, P(kw[k]))
    return f


FOLD_CONVS = [  # N, H, W, Cin, Cout, K, stride, pad, slots, tile
    (2, 10, 10, 128, 64, 1, 1, 0, 4, 0), (3, 8, 8, 128, 128, 3, 1, 1, 2, 5), (2, 12, 12, 64, 128, 3, 2, 1, 1, 6),
    (2, 12, 12, 64, 128, 1, 2, 0, 16, 3), (2, 9, 9, 8, 16, 3, 1, 1, 4, 0), (2, 7, 7, 256, 72, 1, 1, 0, 4, 1),
]


@pytest.mark.parametrize("cfg", FOLD_CONVS)
def test_conv_with_batchnorm_folded_in_the_consumer(cfg):
    """mmvqa_bn_fold: the consuming launch derives the BatchNorm coefficients of its A prologue from the raw sums of
    the producer (train mode) and publishes what mmvqa_bn_coef_fwd / _bwd would have written.  Against torch's
    BatchNorm2d (forward, running statistics with the k-fold rule, backward through the batch statistics) and against
    the same launches fed with precomputed coefficients.  Cin = 8 takes the launcher's fallback (general loaders:
    coefficient launch in front)."""
    N, H, W, Cin, Cout, K, s, p, slots, tile = cfg
    torch.manual_seed(11)
    z1 = (torch.randn(N, Cin, H, W) * 1.5 + 0.3).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(Cin).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    reps = 3
    w = (torch.randn(Cout, Cin, K, K) / math.sqrt(Cin * K * K)).requires_grad_(True)
    y = bn(z1)
    z2 = F.conv2d(torch.relu(y), w, stride=s, padding=p)
    OH, OW = z2.shape[2:]
    for _ in range(reps - 1):
        bn(z1.detach())
    M1 = N * H * W
    sums = torch.stack([z1.detach().double().sum(dim=(0, 2, 3)), (z1.detach().double() ** 2).sum(dim=(0, 2, 3))], 1)
    stat = _spread(sums, slots)
    g, b = bn.weight.detach().to(dev()), bn.bias.detach().to(dev())
    outs = [torch.full((Cin,), float("nan"), device=dev()) for _ in range(4)]
    rm, rv = torch.zeros(Cin, device=dev()), torch.ones(Cin, device=dev())
    nbt = torch.zeros(1, dtype=torch.int64, device=dev())
    xd, wd = nhwc(z1.detach()), w_ohwi(w.detach())
    zo = torch.zeros(N * OH * OW, Cout, device=dev())
    d, _, _ = conv_desc_fwd(xd, wd, N, H, W, Cin, Cout, K, s, p, zo)
    d.a_pro = L.PRO_AFFINE_RELU
    d.a_fold = _fold(stat, slots, 0, 1, M1, g, beta=b, out0=outs[0], out1=outs[1], out2=outs[2], out3=outs[3],
                     run_mean=rm, run_var=rv, nbt=nbt, reps=reps)
    run_igemm(d, L.KIND_FWD, tile=tile)
    assert_close(from_nhwc(zo, N, OH, OW, Cout), z2, TOL, "conv(relu(bn(z1))) with the fold")
    mean = z1.detach().mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(z1.detach().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    assert_close(outs[2], mean, 1e-5, "published mean")
    assert_close(outs[3], invstd, 1e-5, "published invstd")
    assert_close(outs[0], bn.weight.detach() * invstd, 1e-5, "published scale")
    assert_close(rm, bn.running_mean, 1e-5, "running_mean (k-fold)")
    assert_close(rv, bn.running_var, 1e-5, "running_var (k-fold)")
    assert int(nbt) == reps
    # ---- backward of the SECOND BatchNorm (on z2): dz2 = P*G + Q*z2 + R with P, Q, R folded from (sum G, sum G*xhat)
    bn2 = torch.nn.BatchNorm2d(Cout).train()
    with torch.no_grad():
        bn2.weight.uniform_(0.5, 1.5)
    z2d = z2.detach().requires_grad_(True)
    G = torch.randn(N, Cout, OH, OW)
    bn2(z2d).backward(G)
    dz2 = z2d.grad                                   # torch's BatchNorm backward through the batch statistics
    M2 = N * OH * OW
    mean2 = z2.detach().mean(dim=(0, 2, 3))
    invstd2 = 1.0 / torch.sqrt(z2.detach().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    xhat2 = (z2.detach() - mean2[None, :, None, None]) * invstd2[None, :, None, None]
    bs = torch.stack([G.double().sum(dim=(0, 2, 3)), (G * xhat2).double().sum(dim=(0, 2, 3))], 1)
    bstat = _spread(bs, slots)
    g2, m2d, i2d = bn2.weight.detach().to(dev()), mean2.to(dev()), invstd2.to(dev())
    pqr = [torch.full((Cout,), float("nan"), device=dev()) for _ in range(3)]
    dg, db = torch.zeros(Cout, device=dev()), torch.zeros(Cout, device=dev())
    Gd, z2n = nhwc(G), nhwc(z2.detach())
    # reference for the contraction: conv backward fed with torch's dz2
    a1 = torch.relu(y.detach()).requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    F.conv2d(a1, wr, stride=s, padding=p).backward(dz2)
    dx = torch.zeros(N * H * W, Cin, device=dev())
    d = conv_desc_dgrad(Gd, wd, N, H, W, Cin, Cout, K, s, p, dx)
    d.A2, d.a_pro = P(z2n), L.PRO_DZ
    d.a_fold = _fold(bstat, slots, 1, 1, M2, g2, mean=m2d, invstd=i2d, out0=pqr[0], out1=pqr[1], out2=pqr[2], dgamma=dg, dbeta=db)
    run_igemm(d, L.KIND_DGRAD, tile=tile)
    assert_close(from_nhwc(dx, N, H, W, Cin), a1.grad, TOL, "dgrad through the folded BatchNorm backward")
    assert_close(dg, bn2.weight.grad, 1e-5, "dgamma")
    assert_close(db, bn2.bias.grad, 1e-5, "dbeta")
    dw = torch.zeros(Cout, K * K * Cin, device=dev())
    d = conv_desc_wgrad(Gd, xd, N, H, W, Cin, Cout, K, s, p, dw)
    d.A2, d.a_pro = P(z2n), L.PRO_DZ
    d.a_fold = _fold(bstat, slots, 1, 0, M2, g2, mean=m2d, invstd=i2d)
    d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(outs[0]), P(outs[1])
    run_igemm(d, L.KIND_WGRAD, tile=tile if tile != 1 else 0)
    assert_close(dw.view(Cout, K, K, Cin).permute(0, 3, 1, 2), wr.grad, TOL, "wgrad through the folded BatchNorm backward")
    assert_close(dg, bn2.weight.grad, 1e-5, "dgamma untouched by the non-publishing launch")
    # the same two launches on precomputed coefficients (the published P, Q, R) give the same numbers
    dx2 = torch.zeros_like(dx)
    d = conv_desc_dgrad(Gd, wd, N, H, W, Cin, Cout, K, s, p, dx2)
    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z2n), L.PRO_DZ, P(pqr[0]), P(pqr[1]), P(pqr[2])
    run_igemm(d, L.KIND_DGRAD, tile=tile)
    assert_close(dx2, dx, 1e-6, "folded vs precomputed coefficients")


def test_block_end_with_folded_batchnorms():
    """relu(bn3(z3) + bn_d(zd)) and relu(bn3(z3) + x) with the coefficients folded inside the block-end launch"""
    torch.manual_seed(12)
    for N, Cc, H, W, slots, with_d in ((3, 96, 7, 7, 4, True), (2, 256, 9, 9, 2, False), (1, 32, 5, 6, 1, True)):
        z3, zd = torch.randn(N, Cc, H, W) * 2 + 0.5, torch.randn(N, Cc, H, W) * 0.7 - 0.2
        b3, bd = torch.nn.BatchNorm2d(Cc).train(), torch.nn.BatchNorm2d(Cc).train()
        with torch.no_grad():
            for m in (b3, bd):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
        ref = torch.relu(b3(z3) + (bd(zd) if with_d else zd))
        M = N * H * W
        folds, keep = [], []
        for z, m in ((z3, b3), (zd, bd)):
            sums = torch.stack([z.double().sum(dim=(0, 2, 3)), (z.double() ** 2).sum(dim=(0, 2, 3))], 1)
            outs = [torch.zeros(Cc, device=dev()) for _ in range(4)]
            rm, rv = torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())
            nbt = torch.zeros(1, dtype=torch.int64, device=dev())
            st, g, b = _spread(sums, slots), m.weight.detach().to(dev()), m.bias.detach().to(dev())
            keep.append((st, g, b, outs, rm, rv, nbt))
            folds.append(_fold(st, slots, 0, 1, M, g, beta=b, out0=outs[0], out1=outs[1], out2=outs[2], out3=outs[3],
                               run_mean=rm, run_var=rv, nbt=nbt))
        o = torch.zeros(M, Cc, device=dev())
        z3d, zdd = nhwc(z3), nhwc(zd)
        L.check(L.lib().mmvqa_bn_add_relu_fold(L.stream_ptr(), P(z3d), C.byref(folds[0]), P(zdd),
                                               C.byref(folds[1]) if with_d else None, P(o), M, Cc))
        torch.cuda.synchronize()
        assert_close(from_nhwc(o, N, H, W, Cc), ref, TOL, "block end")
        assert_close(keep[0][4], b3.running_mean, 1e-5, "bn3 running mean")
        assert_close(keep[0][5], b3.running_var, 1e-5, "bn3 running var")
        assert int(keep[0][6]) == 1 and int(keep[1][6]) == (1 if with_d else 0)
        if with_d:
            assert_close(keep[1][5], bd.running_var, 1e-5, "downsample bn running var")
            assert_close(keep[1][3][0], bd.weight.detach() / torch.sqrt(zd.var(dim=(0, 2, 3), unbiased=False) + 1e-5), 1e-5, "bn_d scale")


def test_maxpool():
    torch.manual_seed(6)
    N, Cc, H, W = 2, 8, 11, 12
    z = torch.randn(N, Cc, H, W)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.2
    a = torch.relu(z * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    p_ref = F.max_pool2d(a, 3, 2, 1)
    OH, OW = p_ref.shape[2:]
    gp = torch.randn_like(p_ref)
    p_ref.backward(gp)
    zd, scd, shd = nhwc(z), sc.to(dev()), sh.to(dev())
    out = torch.zeros(N * OH * OW, Cc, device=dev())
    idx = torch.zeros(N * OH * OW * Cc, dtype=torch.uint8, device=dev())
    L.check(L.lib().mmvqa_maxpool_fwd(L.stream_ptr(), P(zd), P(scd), P(shd), P(out), P(idx), N, H, W, Cc, OH, OW))
    assert_close(from_nhwc(out, N, OH, OW, Cc), p_ref, 1e-6, "maxpool fwd")
    extra = torch.randn(N, Cc, H, W)
    mu, istd = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    g0 = torch.zeros(N * H * W, Cc, device=dev())
    stat = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=dev())
    gpd, exd, mud, isd = nhwc(gp), nhwc(extra), mu.to(dev()), istd.to(dev())  # keep alive across the async launch
    L.check(L.lib().mmvqa_maxpool_bwd(L.stream_ptr(), P(gpd), P(idx), P(exd), P(zd), P(scd), P(shd),
                                      P(mud), P(isd), P(g0), P(stat), N, H, W, Cc, OH, OW))
    torch.cuda.synchronize()
    mask = (a.detach() > 0).float()
    g_ref = (a.grad + extra) * mask
    assert_close(from_nhwc(g0, N, H, W, Cc), g_ref, 1e-5, "maxpool bwd")
    xh = (z - mu[None, :, None, None]) * istd[None, :, None, None]
    st = stat.sum(0).cpu()
    assert_close(st[:, 0], g_ref.sum(dim=(0, 2, 3)).double(), 1e-5, "sum g")
    assert_close(st[:, 1], (g_ref * xh).sum(dim=(0, 2, 3)).double(), 1e-5, "sum g xhat")


# ----------------------------------------------------------------------------- LayerNorm / embeddings
@pytest.mark.parametrize("rows,H,eps", [(37, 96, 1e-12), (512, 768, 1e-12), (64, 768, 1e-5)])
def test_layernorm(rows, H, eps):
    torch.manual_seed(7)
    x = (torch.randn(rows, H) * 2 + 0.5).requires_grad_(True)
    r = torch.randn(rows, H).requires_grad_(True)
    g, b = (torch.rand(H) + 0.5).requires_grad_(True), torch.randn(H).requires_grad_(True)
    y_ref = F.layer_norm(x + r, (H,), g, b, eps)
    dy, dres = torch.randn(rows, H), torch.randn(rows, H)
    y_ref.backward(dy)
    xd, rd, gd, bd = (t.detach().to(dev()) for t in (x, r, g, b))
    y, s = torch.zeros(rows, H, device=dev()), torch.zeros(rows, H, device=dev())
    mean, rstd = torch.zeros(rows, device=dev()), torch.zeros(rows, device=dev())
    L.check(L.lib().mmvqa_layernorm_fwd(L.stream_ptr(), P(xd), P(rd), P(gd), P(bd), P(y), P(s), P(mean), P(rstd), rows, H, eps))
    assert_close(y, y_ref, TOL, "ln y")
    dx = torch.zeros(rows, H, device=dev())
    dg, db = torch.zeros(H, device=dev()), torch.zeros(H, device=dev())
    dyd, dresd = dy.to(dev()), dres.to(dev())
    L.check(L.lib().mmvqa_layernorm_bwd(L.stream_ptr(), P(dyd), P(s), P(gd), P(mean), P(rstd), P(dresd),
                                        P(dx), P(dg), P(db), rows, H))
    torch.cuda.synchronize()
    assert_close(dx, x.grad + dres, TOL, "ln dx")
    assert_close(dg, g.grad, TOL, "ln dgamma")
    assert_close(db, b.grad, TOL, "ln dbeta")


def test_embed():
    torch.manual_seed(8)
    B, T, H, V, nv = 3, 12, 96, 40, 5
    emb = O.OracleBertEmbeddings(V, H, 32).eval()
    ids = torch.randint(0, V, (B, T))
    ids[:, 1:6] = 0
    seg = torch.randint(0, 2, (B, T))
    vis = torch.randn(nv, B, H, requires_grad=True)
    h = emb(ids, seg).clone()
    for n in range(nv):
        h[:, n, :] = vis[n]
    dh = torch.randn(B, T, H)
    h.backward(dh)
    sd = {k: v.detach().to(dev()) for k, v in emb.state_dict().items()}
    out, xh = torch.zeros(B * T, H, device=dev()), torch.zeros(B * T, H, device=dev())
    rstd = torch.zeros(B * T, device=dev())
    idsd, segd, visd = ids.to(dev()), seg.to(dev()), vis.detach().to(dev())
    L.check(L.lib().mmvqa_embed_fwd(L.stream_ptr(), P(idsd), P(segd), P(sd["word_embeddings.weight"]),
                                    P(sd["position_embeddings.weight"]), P(sd["token_type_embeddings.weight"]),
                                    P(sd["LayerNorm.weight"]), P(sd["LayerNorm.bias"]), P(visd), P(out), P(xh), P(rstd),
                                    B, T, H, nv, 1e-12, 0.0, 0))
    assert_close(out.view(B, T, H), h, TOL, "embed fwd")
    dw, dp, dt = (torch.zeros_like(sd[k]) for k in ("word_embeddings.weight", "position_embeddings.weight",
                                                     "token_type_embeddings.weight"))
    dg, db = torch.zeros(H, device=dev()), torch.zeros(H, device=dev())
    dvis = torch.zeros(nv, B, H, device=dev())
    dhd = dh.to(dev())
    L.check(L.lib().mmvqa_embed_bwd(L.stream_ptr(), P(dhd), P(idsd), P(segd), P(xh), P(rstd),
                                    P(sd["LayerNorm.weight"]), P(dw), P(dp), P(dt), P(dg), P(db), P(dvis), B, T, H, nv,
                                    0.0, 0, 0))
    torch.cuda.synchronize()
    assert_close(dvis, vis.grad, 1e-6, "dvis")
    assert_close(dw, emb.word_embeddings.weight.grad, TOL, "dword")
    assert torch.all(dw[0] == 0)  # padding_idx row
    assert_close(dp, emb.position_embeddings.weight.grad, TOL, "dpos")
    assert_close(dt, emb.token_type_embeddings.weight.grad, TOL, "dtype")
    assert_close(dg, emb.LayerNorm.weight.grad, TOL, "dgamma")
    assert_close(db, emb.LayerNorm.bias.grad, TOL, "dbeta")


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,T,heads,D", [(2, 32, 12, 64), (3, 10, 12, 8), (2, 28, 12, 64), (2, 40, 4, 64),
                                         (1, 75, 2, 64)])
def test_attention_bert(B, T, heads, D):
    """models/transformer.py:19-30 (key-axis mask), ragged masks"""
    torch.manual_seed(9)
    H = heads * D
    qkv = torch.randn(B * T, 3 * H, requires_grad=True)
    mask = torch.ones(B, T, dtype=torch.long)
    for b in range(B):
        mask[b, T - 2 * b - 1:] = 0
    q, k, v = (qkv[:, i * H:(i + 1) * H].view(B, T, heads, D).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(-2, -1) / float(math.sqrt(D))
    sc = sc - 10000.0 * (1.0 - mask[:, None, None, :].float())
    pr = F.softmax(sc, dim=-1)
    ctx_ref = (pr @ v).transpose(1, 2).contiguous().view(B * T, H)
    dctx = torch.randn(B * T, H)
    ctx_ref.backward(dctx)
    qd, md = qkv.detach().to(dev()), mask.to(dev())
    ctx = torch.zeros(B * T, H, device=dev())
    probs = torch.zeros(B, heads, T, T, device=dev())
    a = L.AttnDesc()
    a.q, a.k, a.v = P(qd), P(qd) + 4 * H, P(qd) + 8 * H
    a.row_stride, a.head_stride = 3 * H, D
    a.out, a.out_row_stride, a.out_head_stride = P(ctx), H, D
    a.mask, a.mask_on_query, a.probs = P(md), 0, P(probs)
    a.B, a.T, a.heads, a.sqrt_d = B, T, heads, math.sqrt(D)
    L.check(L.lib().mmvqa_attention(C.byref(a), D, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert_close(ctx, ctx_ref, TOL, "ctx")
    assert_close(probs.transpose(-1, -2), pr, TOL, "probs")
    dqkv = torch.zeros(B * T, 3 * H, device=dev())
    dctxd = dctx.to(dev())
    a.dout, a.dq, a.dk, a.dv = P(dctxd), P(dqkv), P(dqkv) + 4 * H, P(dqkv) + 8 * H
    L.check(L.lib().mmvqa_attention(C.byref(a), D, 1, L.stream_ptr()))
    torch.cuda.synchronize()
    assert_close(dqkv, qkv.grad, TOL, "dqkv")


@pytest.mark.parametrize("B,T,heads", [(16, 32, 12), (3, 28, 12), (2, 10, 2), (1, 1, 1)])
def test_fused_qkv_projection_and_attention(B, T, heads):
    """mmvqa_qkv_attention_fwd: proj_q / proj_k / proj_v + attention of a BertLayer in one launch
    (models/transformer.py:19-30), ragged key masks, T < 32; q|k|v, probabilities and context against torch, and -- with
    dropout -- against the two-launch path it replaces (same counter-based stream: bit-equal masks)."""
    torch.manual_seed(31)
    D = 64
    H = heads * D
    xn = torch.randn(B * T, H)
    W = torch.randn(3 * H, H) / math.sqrt(H)
    bias = torch.randn(3 * H) * 0.1
    mask = torch.ones(B, T, dtype=torch.long)
    for b in range(B):
        mask[b, max(1, T - 2 * (b % 5) - 1):] = 0
    qkv_ref = xn @ W.t() + bias
    q, k, v = (qkv_ref[:, i * H:(i + 1) * H].view(B, T, heads, D).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(-2, -1) / float(math.sqrt(D)) - 10000.0 * (1.0 - mask[:, None, None, :].float())
    pr = F.softmax(sc, dim=-1)
    ctx_ref = (pr @ v).transpose(1, 2).contiguous().view(B * T, H)
    xd, Wd, bd, md = xn.to(dev()), W.to(dev()), bias.to(dev()), mask.to(dev())

    def fused(drop_p, seed):
        qkv = torch.full((B * T, 3 * H), float("nan"), device=dev())
        probs = torch.zeros(B, heads, T, T, device=dev())
        ctx = torch.full((B * T, H), float("nan"), device=dev())
        L.check(L.lib().mmvqa_qkv_attention_fwd(L.stream_ptr(), P(xd), P(Wd), P(bd), P(md), P(qkv), P(probs), P(ctx), B, T, H,
                                                heads, drop_p, seed))
        torch.cuda.synchronize()
        return qkv, probs, ctx

    qkv, probs, ctx = fused(0.0, 0)
    assert_close(qkv, qkv_ref, TOL, "q|k|v")
    assert_close(probs.transpose(-1, -2), pr, TOL, "probs")
    assert_close(ctx, ctx_ref, TOL, "ctx")
    # dropout: the two-launch path (projection values from the fused run, so that only the attention differs)
    qkv, probs, ctx = fused(0.3, 1234)
    ctx2 = torch.zeros(B * T, H, device=dev())
    probs2 = torch.zeros(B, heads, T, T, device=dev())
    a = L.AttnDesc()
    a.q, a.k, a.v = P(qkv), P(qkv) + 4 * H, P(qkv) + 8 * H
    a.row_stride, a.head_stride = 3 * H, D
    a.out, a.out_row_stride, a.out_head_stride = P(ctx2), H, D
    a.mask, a.mask_on_query, a.probs = P(md), 0, P(probs2)
    a.B, a.T, a.heads, a.sqrt_d, a.drop_p, a.seed = B, T, heads, math.sqrt(D), 0.3, 1234
    L.check(L.lib().mmvqa_attention(C.byref(a), D, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert_close(probs, probs2, 1e-6, "probabilities, fused vs two launches")
    assert_close(ctx, ctx2, 1e-5, "context with dropout, fused vs two launches")
    assert relerr(ctx, ctx_ref) > 1e-2      # dropout really was applied


@pytest.mark.parametrize("B,T,es", [(2, 32, 96), (3, 9, 12), (2, 40, 96)])
def test_attention_realformer(B, T, es):
    """models/realformer.py:30-45: k,q,v split, residual scores, QUERY-axis mask, prev chain"""
    torch.manual_seed(10)
    h = 8
    kqv = torch.randn(B * T * h, 3 * es, requires_grad=True)
    prev = (torch.randn(B, T, T, h) * 0.5).requires_grad_(True)
    mask = torch.ones(B, T, dtype=torch.long)
    for b in range(B):
        mask[b, T - b - 1:] = 0
    x = kqv.view(B, T, h, 3 * es)
    k, q, v = torch.split(x, es, dim=-1)
    att = torch.einsum("bihk,bjhk->bijh", q, k) / es ** 0.5 + prev
    att = att - 10000.0 * (1.0 - mask.unsqueeze(-1).unsqueeze(-1).expand(att.size()).float())
    pr = F.softmax(att, dim=2)
    res_ref = torch.einsum("btih,bihs->bths", pr, v).reshape(B * T, h * es)
    dres, dprev_next = torch.randn(B * T, h * es), torch.randn(B, T, T, h) * 0.1
    (res_ref * dres).sum().backward(retain_graph=True)
    g_kqv_1, g_prev_1 = kqv.grad.clone(), prev.grad.clone()
    kqv.grad = None
    prev.grad = None
    ((res_ref * dres).sum() + (att * dprev_next).sum()).backward()
    kd, md, pd = kqv.detach().to(dev()), mask.to(dev()), prev.detach().to(dev())
    res = torch.zeros(B * T, h * es, device=dev())
    probs = torch.zeros(B, h, T, T, device=dev())
    prev_out = torch.zeros(B, T, T, h, device=dev())
    a = L.AttnDesc()
    a.k, a.q, a.v = P(kd), P(kd) + 4 * es, P(kd) + 8 * es
    a.row_stride, a.head_stride = h * 3 * es, 3 * es
    a.out, a.out_row_stride, a.out_head_stride = P(res), h * es, es
    a.mask, a.mask_on_query, a.probs = P(md), 1, P(probs)
    a.prev_in, a.prev_out = P(pd), P(prev_out)
    a.B, a.T, a.heads, a.sqrt_d = B, T, h, es ** 0.5
    L.check(L.lib().mmvqa_attention(C.byref(a), es, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert_close(prev_out, att, 1e-6, "prev_out")
    assert_close(res, res_ref, 3e-4, "res")   # 96-deep dot products + 40-key softmax in a different fp32 order
    dk = torch.zeros(B * T * h, 3 * es, device=dev())
    dpo = torch.zeros(B, T, T, h, device=dev())
    dresd, dpn = dres.to(dev()), dprev_next.to(dev())
    a.dout, a.dk, a.dq, a.dv = P(dresd), P(dk), P(dk) + 4 * es, P(dk) + 8 * es
    a.dprev_in, a.dprev_out = P(dpn), P(dpo)
    L.check(L.lib().mmvqa_attention(C.byref(a), es, 1, L.stream_ptr()))
    torch.cuda.synchronize()
    assert_close(dk, kqv.grad, 5e-4, "dkqv")
    assert_close(dpo, prev.grad, 5e-4, "dprev")


# ----------------------------------------------------------------------------- losses / optimizer
def test_mlm_loss(golden_dir):
    torch.manual_seed(11)
    B, T, V = 4, 16, 30522
    logits = (torch.randn(B, T, V) * 2).requires_grad_(True)
    tgt = torch.zeros(B, T, dtype=torch.long)
    tgt[0, 3], tgt[1, 7], tgt[2, 9], tgt[3, 1] = 17, 30521, 5, 1234
    loss_ref, lp = O.mlm_loss(logits, tgt)
    loss_ref.backward()
    pred_ref, nc, nm = O.mlm_accuracy(lp, tgt)
    lg = logits.detach().to(dev()).requires_grad_(True)
    loss, pred, stats = mmvqa_amd.mlm_loss(lg, tgt.to(dev()))
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    assert_close(lg.grad, logits.grad, TOL, "dlogits")
    assert torch.equal(pred.cpu()[tgt > 0], pred_ref)           # index ops bit-exact
    assert torch.equal(pred.cpu(), lp.argmax(-1))
    assert int(stats[1]) == nm and int(stats[2]) == nc


def test_asl_supcon_golden(golden_dir):
    import numpy as np
    g = dict(np.load(f"{golden_dir}/losses.npz"))
    lg = torch.from_numpy(g["asl_logits"]).to(dev()).requires_grad_(True)
    l = mmvqa_amd.asl_loss(lg, torch.from_numpy(g["asl_target"]).to(dev()))
    l.backward()
    assert abs(float(l) - float(g["asl"])) < 2e-5 * abs(float(g["asl"]))
    assert_close(lg.grad, torch.from_numpy(g["asl_dlogits"]), TOL, "asl grad")
    f = torch.from_numpy(g["sc_feat"]).to(dev()).requires_grad_(True)
    l = mmvqa_amd.supcon_loss(f)
    l.backward()
    assert abs(float(l) - float(g["sc"])) < 2e-5 * abs(float(g["sc"]))
    assert_close(f.grad, torch.from_numpy(g["sc_dfeat"]), TOL, "supcon grad")
    lg = torch.from_numpy(g["mlm_logits"]).to(dev()).requires_grad_(True)
    l, pred, _ = mmvqa_amd.mlm_loss(lg, torch.from_numpy(g["mlm_target"]).to(dev()))
    l.backward()
    assert abs(float(l) - float(g["mlm"])) < 2e-5 * abs(float(g["mlm"]))
    assert_close(lg.grad, torch.from_numpy(g["mlm_dlogits"]), TOL, "mlm grad")
    t = torch.from_numpy(g["mlm_target"])
    assert np.array_equal(pred.cpu()[t > 0].numpy(), g["mlm_pred"])


@pytest.mark.parametrize("N,D", [(5, 16), (16, 128), (37, 128), (64, 128), (128, 128), (192, 128), (320, 96)])
def test_supcon_sizes(N, D):
    """SupCon / SimCLR over 2N views for N up to the all-gathered set of an 8-GPU job and beyond (2N = 32*8 = 256
    for BASELINE configs[3]; 48*8 = 384 for the README's batch-48 runs; ragged row/column tiles): loss and
    gradient vs the oracle (models/SupConLoss/loss.py:57-96)"""
    torch.manual_seed(15 + N)
    f = F.normalize(torch.randn(N, 2, D), dim=2).requires_grad_(True)
    ref = O.supcon_simclr(f)
    ref.backward()
    fd = f.detach().to(dev()).requires_grad_(True)
    l = mmvqa_amd.supcon_loss(fd)
    l.backward()
    assert abs(float(l) - float(ref)) <= 2e-5 * abs(float(ref)), (float(l), float(ref))
    assert_close(fd.grad, f.grad, TOL, f"supcon grad N={N}")
    # un-normalised features (the ABI does not assume unit rows): large scores, max taken on the diagonal
    g = (torch.randn(N, 2, D) * 0.1).requires_grad_(True)
    ref = O.supcon_simclr(g)
    ref.backward()
    gd = g.detach().to(dev()).requires_grad_(True)
    l = mmvqa_amd.supcon_loss(gd)
    l.backward()
    assert abs(float(l) - float(ref)) <= 5e-5 * abs(float(ref)), (float(l), float(ref))
    assert_close(gd.grad, g.grad, 2e-4, f"supcon grad (unnormalised) N={N}")


def test_mlm_loss_unaligned_and_upstream_scale():
    """rows that are not 16-byte aligned take the scalar kernel; the upstream gradient is applied on the device"""
    torch.manual_seed(16)
    for V in (50, 1001, 30522):
        lg = (torch.randn(3, 5, V) * 2).requires_grad_(True)
        tgt = torch.randint(0, V, (3, 5))
        (O.mlm_loss(lg, tgt)[0] * 0.37).backward()
        x = lg.detach().to(dev()).requires_grad_(True)
        loss, pred, stats = mmvqa_amd.mlm_loss(x, tgt.to(dev()))
        (loss * 0.37).backward()
        assert_close(x.grad, lg.grad, TOL, f"scaled dlogits V={V}")
        assert torch.equal(pred.cpu(), lg.detach().log_softmax(-1).argmax(-1))


def test_adam():
    torch.manual_seed(12)
    n = 4096 + 8
    p, g = torch.randn(n), torch.randn(n)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = (t.clone().to(dev()) for t in (p, g, m, v))
    for step in (1, 2, 3):
        O.adam_step(p, g, m, v, step, 2e-5)
        L.check(L.lib().mmvqa_adam(L.stream_ptr(), P(pd), P(gd), P(md), P(vd), n, 2e-5, 0.9, 0.999, 1e-8, step, 1.0, 0))
    torch.cuda.synchronize()
    assert_close(pd, p, 1e-6, "adam p")
    assert_close(vd, v, 2e-6, "adam v")
    ref = torch.optim.Adam([torch.nn.Parameter(torch.ones(4))], lr=2e-5)  # same defaults as roco_train.py:90
    assert ref.defaults["betas"] == (0.9, 0.999) and ref.defaults["eps"] == 1e-8


def test_meanpool_l2norm():
    torch.manual_seed(13)
    B, T, H = 3, 11, 96
    h = torch.randn(B, T, H, requires_grad=True)
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, 6:] = 0
    mask[2, :] = 0
    ref = O.mean_pooling(h, mask)
    dp = torch.randn(B, H)
    ref.backward(dp)
    hd, md = h.detach().to(dev()), mask.to(dev())
    out = torch.zeros(B, H, device=dev())
    L.check(L.lib().mmvqa_meanpool_fwd(L.stream_ptr(), P(hd), P(md), P(out), B, T, H))
    assert_close(out, ref, TOL, "meanpool")
    dh = torch.zeros(B, T, H, device=dev())
    dpd = dp.to(dev())
    L.check(L.lib().mmvqa_meanpool_bwd(L.stream_ptr(), P(dpd), P(md), P(dh), B, T, H, 0))
    torch.cuda.synchronize()
    assert_close(dh, h.grad, TOL, "meanpool bwd")
    x = torch.randn(5, 128, requires_grad=True)
    y_ref = F.normalize(x, dim=1)
    dy = torch.randn(5, 128)
    y_ref.backward(dy)
    xd = x.detach().to(dev())
    y, nrm, dx = torch.zeros(5, 128, device=dev()), torch.zeros(5, device=dev()), torch.zeros(5, 128, device=dev())
    L.check(L.lib().mmvqa_l2norm_fwd(L.stream_ptr(), P(xd), P(y), P(nrm), 5, 128))
    dyd = dy.to(dev())
    L.check(L.lib().mmvqa_l2norm_bwd(L.stream_ptr(), P(dyd), P(y), P(nrm), P(dx), 5, 128))
    torch.cuda.synchronize()
    assert_close(y, y_ref, TOL, "l2norm")
    assert_close(dx, x.grad, TOL, "l2norm bwd")


# ----------------------------------------------------------------------------- EfficientNetV2 pieces (timm MBConv)
def _same_pad(size, k, s):
    out = math.ceil(size / s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


@pytest.mark.parametrize("N,H,W,Cc,stride", [(2, 14, 14, 64, 1), (3, 7, 7, 96, 1), (2, 14, 14, 32, 2), (2, 28, 28, 64, 2),
                                           (1, 40, 36, 32, 1), (2, 9, 11, 64, 2)])
def test_dwconv_fwd_bwd(N, H, W, Cc, stride):
    """depthwise 3x3 with TF-SAME padding, BN+SiLU applied on load, statistics, data and weight gradients
    (image-tiled kernels where the map fits in LDS, pixel-strided ones for 40x36) vs torch conv2d(groups=C)"""
    torch.manual_seed(20)
    z1 = torch.randn(N, Cc, H, W)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.3
    w = (torch.randn(Cc, 1, 3, 3) / 3).requires_grad_(True)
    pt, pb = _same_pad(H, 3, stride)
    pl, pr = _same_pad(W, 3, stride)
    zin = z1.clone().requires_grad_(True)
    a = F.silu(zin * sc[None, :, None, None] + sh[None, :, None, None])
    z2 = F.conv2d(F.pad(a, (pl, pr, pt, pb)), w, stride=stride, groups=Cc)
    OH, OW = z2.shape[2:]
    z1d, wd = nhwc(z1), w.detach().reshape(Cc, 9).contiguous().to(dev())
    scd, shd = sc.to(dev()), sh.to(dev())
    out = torch.zeros(N * OH * OW, Cc, device=dev())
    stat = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=dev())
    L.check(L.lib().mmvqa_dwconv_fwd(L.stream_ptr(), P(z1d), P(scd), P(shd), P(wd), P(out), P(stat), N, H, W, Cc, OH, OW, stride, pt))
    torch.cuda.synchronize()
    assert pt == pl
    assert_close(from_nhwc(out, N, OH, OW, Cc), z2, TOL, "dwconv fwd")
    st = stat.sum(0).cpu()
    assert_close(st[:, 0], z2.sum(dim=(0, 2, 3)).double(), 1e-5, "sum")
    assert_close(st[:, 1], (z2.double() ** 2).sum(dim=(0, 2, 3)), 1e-5, "sumsq")
    # backward through dz2 = Pc*g2 + Qc*z2 + Rc
    g2 = torch.randn_like(z2)
    Pc, Qc, Rc = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.1, torch.randn(Cc) * 0.1
    dz2 = g2 * Pc[None, :, None, None] + z2.detach() * Qc[None, :, None, None] + Rc[None, :, None, None]
    z2.backward(dz2)
    mu, istd = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    g2d = nhwc(g2)
    coef = [t.to(dev()) for t in (Pc, Qc, Rc, mu, istd)]
    g1 = torch.zeros(N * H * W, Cc, device=dev())
    bst = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=dev())
    L.check(L.lib().mmvqa_dwconv_bwd_data(L.stream_ptr(), P(g2d), P(out), P(coef[0]), P(coef[1]), P(coef[2]), P(wd), P(z1d), P(scd),
                                          P(shd), P(coef[3]), P(coef[4]), P(g1), P(bst), N, H, W, Cc, OH, OW, stride, pt))
    torch.cuda.synchronize()
    # zin.grad = dL/dz1 = da * silu'(.) * sc ; the kernel returns du = da * silu'(.) (the BN scale is applied by the next stage)
    du_ref = zin.grad / sc[None, :, None, None]
    assert_close(from_nhwc(g1, N, H, W, Cc), du_ref, TOL, "dwconv bwd data")
    xhat = (z1 - mu[None, :, None, None]) * istd[None, :, None, None]
    bs = bst.sum(0).cpu()
    assert_close(bs[:, 0], du_ref.sum(dim=(0, 2, 3)).double(), 1e-4, "sum du")
    assert_close(bs[:, 1], (du_ref * xhat).sum(dim=(0, 2, 3)).double(), 1e-4, "sum du xhat")
    dw = torch.zeros(Cc, 9, device=dev())
    L.check(L.lib().mmvqa_dwconv_bwd_weight(L.stream_ptr(), P(g2d), P(out), P(coef[0]), P(coef[1]), P(coef[2]), P(z1d), P(scd), P(shd),
                                            P(dw), N, H, W, Cc, OH, OW, stride, pt))
    torch.cuda.synchronize()
    assert_close(dw.view(Cc, 1, 3, 3), w.grad, TOL, "dwconv bwd weight")


@pytest.mark.parametrize("N,H,W,Cc,stride,slots", [(3, 14, 14, 64, 1, 16), (2, 9, 11, 96, 2, 4), (2, 40, 36, 32, 1, 1)])
def test_elementwise_consumers_with_folded_batchnorm(N, H, W, Cc, stride, slots):
    """mmvqa_*_fold (EfficientNetV2 path, round 3): the depthwise convolution (forward: BatchNorm of its input; data /
    weight gradient: backward coefficients of its output's BatchNorm), the squeeze-excite pooling and the block end derive
    the coefficients from raw sums inside the launch.  Same results as the plain entries on coefficients computed in
    float64 here; published coefficients, running statistics (k-fold rule, reps = 2), batch counter, dgamma / dbeta
    against the formulas of mmvqa_bn_coef_fwd / _bwd; tile and pixel-strided kernels (40x36 does not fit in LDS)."""
    torch.manual_seed(33)
    z1 = torch.randn(N, Cc, H, W) * 1.5 + 0.3
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.3
    M = N * H * W
    sums = torch.stack([z1.double().sum(dim=(0, 2, 3)), (z1.double() ** 2).sum(dim=(0, 2, 3))], 1)
    mean = sums[:, 0] / M
    var = sums[:, 1] / M - mean * mean
    sc = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    sh = (beta.double() - mean * sc.double()).float()
    w = torch.randn(Cc, 9) / 3
    pt, _ = _same_pad(H, 3, stride)
    OH, OW = -(-H // stride), -(-W // stride)
    z1d, wd, scd, shd, gd, bd = nhwc(z1), w.to(dev()), sc.to(dev()), sh.to(dev()), gamma.to(dev()), beta.to(dev())

    def fwd_fold(publish=1):
        outs = [torch.zeros(Cc, device=dev()) for _ in range(4)]
        rm, rv = torch.full((Cc,), 0.25, device=dev()), torch.full((Cc,), 2.0, device=dev())
        nbt = torch.zeros(1, dtype=torch.int64, device=dev())
        st = _spread(sums, slots)
        f = _fold(st, slots, 0, publish, M, gd, beta=bd, out0=outs[0], out1=outs[1], out2=outs[2], out3=outs[3], run_mean=rm,
                  run_var=rv, nbt=nbt, reps=2)
        return f, (st, outs, rm, rv, nbt)

    def check_published(keep, what):
        st, outs, rm, rv, nbt = keep
        assert_close(outs[0], sc, 1e-6, what + ": scale")
        assert_close(outs[1], sh, 1e-5, what + ": shift")
        assert_close(outs[3], (1.0 / torch.sqrt(var + 1e-5)).float(), 1e-6, what + ": invstd")
        unb = var * M / (M - 1)
        assert_close(rm, (0.81 * 0.25 + 0.19 * mean).float(), 1e-5, what + ": running mean (2 updates)")
        assert_close(rv, (0.81 * 2.0 + 0.19 * unb).float(), 1e-5, what + ": running var (2 updates)")
        assert int(nbt) == 2, what

    # depthwise forward
    lib = L.lib()
    outp, outf = torch.zeros(N * OH * OW, Cc, device=dev()), torch.zeros(N * OH * OW, Cc, device=dev())
    stp, stf = (torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=dev()) for _ in range(2))
    L.check(lib.mmvqa_dwconv_fwd(L.stream_ptr(), P(z1d), P(scd), P(shd), P(wd), P(outp), P(stp), N, H, W, Cc, OH, OW, stride, pt))
    f, keep = fwd_fold()
    L.check(lib.mmvqa_dwconv_fwd_fold(L.stream_ptr(), P(z1d), None, None, P(wd), P(outf), P(stf), N, H, W, Cc, OH, OW, stride, pt,
                                      C.byref(f)))
    torch.cuda.synchronize()
    assert_close(outf, outp, 2e-6, "dwconv forward: fold vs coefficient arrays")
    assert_close(stf.sum(0), stp.sum(0), 1e-6, "dwconv forward statistics")
    check_published(keep, "dwconv forward")
    # squeeze-excite pooling (a fold that does not publish leaves the outputs alone)
    poolp, poolf = torch.zeros(N, Cc, device=dev()), torch.zeros(N, Cc, device=dev())
    L.check(lib.mmvqa_se_pool(L.stream_ptr(), P(z1d), P(scd), P(shd), P(poolp), N, H * W, Cc))
    f, keep = fwd_fold(publish=0)
    L.check(lib.mmvqa_se_pool_fold(L.stream_ptr(), P(z1d), None, None, P(poolf), N, H * W, Cc, C.byref(f)))
    torch.cuda.synchronize()
    assert_close(poolf, poolp, 2e-6, "se_pool: fold vs coefficient arrays")
    assert float(keep[1][0].abs().max()) == 0.0 and int(keep[4]) == 0, "a non-publishing fold wrote something"
    f, keep = fwd_fold()
    L.check(lib.mmvqa_se_pool_fold(L.stream_ptr(), P(z1d), None, None, P(poolf), N, H * W, Cc, C.byref(f)))
    torch.cuda.synchronize()
    check_published(keep, "se_pool")
    # block end: silu(bn(z)) + idn, and bn(z) alone
    idn = torch.randn(M, Cc, device=dev())
    for pre, with_idn in ((L.ACT_SILU, True), (L.ACT_NONE, True), (L.ACT_NONE, False)):
        op, of = torch.zeros(M, Cc, device=dev()), torch.zeros(M, Cc, device=dev())
        L.check(lib.mmvqa_bn_act_add(L.stream_ptr(), P(z1d), P(scd), P(shd), pre, P(idn) if with_idn else None, None, None, 0,
                                     L.ACT_NONE, P(op), M, Cc))
        f, keep = fwd_fold()
        L.check(lib.mmvqa_bn_act_add_fold(L.stream_ptr(), P(z1d), C.byref(f), pre, P(idn) if with_idn else None, None, L.ACT_NONE,
                                          P(of), M, Cc))
        torch.cuda.synchronize()
        assert_close(of, op, 2e-6, f"block end (pre {pre}, idn {with_idn}): fold vs coefficient arrays")
        check_published(keep, "block end")
    # backward coefficients of the depthwise output's BatchNorm: dz2 = P*g2 + Q*z2 + R from (sum g, sum g*xhat)
    Mo = N * OH * OW
    g2 = torch.randn(Mo, Cc, device=dev())
    mu2, is2, gam2 = torch.randn(Cc) * 0.2, torch.rand(Cc) + 0.5, torch.rand(Cc) + 0.5
    bsum = torch.stack([torch.randn(Cc).double() * 5, torch.randn(Cc).double() * 5], 1)
    p = gam2.double() * is2.double()
    c1, c2 = bsum[:, 0] / Mo, bsum[:, 1] / Mo
    Pc, Qc, Rc = p.float(), (-p * c2 * is2.double()).float(), (p * (c2 * is2.double() * mu2.double() - c1)).float()
    coef = [t.to(dev()) for t in (Pc, Qc, Rc, mu2, is2, gam2)]
    mu1, is1 = (t.to(dev()) for t in (mean.float(), (1.0 / torch.sqrt(var + 1e-5)).float()))
    res = {}
    for name in ("plain", "fold"):
        g1 = torch.zeros(M, Cc, device=dev())
        bst = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=dev())
        dw = torch.zeros(Cc, 9, device=dev())
        if name == "plain":
            L.check(lib.mmvqa_dwconv_bwd_data(L.stream_ptr(), P(g2), P(outp), P(coef[0]), P(coef[1]), P(coef[2]), P(wd), P(z1d), P(scd),
                                              P(shd), P(mu1), P(is1), P(g1), P(bst), N, H, W, Cc, OH, OW, stride, pt))
            L.check(lib.mmvqa_dwconv_bwd_weight(L.stream_ptr(), P(g2), P(outp), P(coef[0]), P(coef[1]), P(coef[2]), P(z1d), P(scd), P(shd),
                                                P(dw), N, H, W, Cc, OH, OW, stride, pt))
        else:
            st = _spread(bsum, slots)
            pub = [torch.zeros(Cc, device=dev()) for _ in range(3)]
            dg, db = torch.full((Cc,), 1.5, device=dev()), torch.full((Cc,), -0.5, device=dev())
            fd = _fold(st, slots, 1, 1, Mo, coef[5], mean=coef[3], invstd=coef[4], out0=pub[0], out1=pub[1], out2=pub[2], dgamma=dg, dbeta=db)
            fw = _fold(st, slots, 1, 0, Mo, coef[5], mean=coef[3], invstd=coef[4])
            L.check(lib.mmvqa_dwconv_bwd_data_fold(L.stream_ptr(), P(g2), P(outp), None, None, None, P(wd), P(z1d), P(scd), P(shd),
                                                   P(mu1), P(is1), P(g1), P(bst), N, H, W, Cc, OH, OW, stride, pt, C.byref(fd)))
            L.check(lib.mmvqa_dwconv_bwd_weight_fold(L.stream_ptr(), P(g2), P(outp), None, None, None, P(z1d), P(scd), P(shd), P(dw),
                                                     N, H, W, Cc, OH, OW, stride, pt, C.byref(fw)))
            torch.cuda.synchronize()
            for got, want, nm in zip(pub, (Pc, Qc, Rc), "PQR"):
                assert_close(got, want, 1e-5, "published " + nm)
            assert_close(dg, 1.5 + bsum[:, 1].float(), 1e-5, "dgamma += sum g*xhat")
            assert_close(db, -0.5 + bsum[:, 0].float(), 1e-5, "dbeta += sum g")
        torch.cuda.synchronize()
        res[name] = (g1, bst.sum(0), dw)
    assert_close(res["fold"][0], res["plain"][0], 1e-5, "dwconv data gradient: fold vs coefficient arrays")
    assert_close(res["fold"][1], res["plain"][1], 1e-5, "dwconv data gradient statistics")
    assert_close(res["fold"][2], res["plain"][2], 1e-5, "dwconv weight gradient: fold vs coefficient arrays")


def test_squeeze_excite_and_block_end_kernels():
    """se_pool / se_dgate / act_bwd_stats / bn_act_add vs plain torch (C not a multiple of 64, HW not a multiple of 16)"""
    torch.manual_seed(21)
    N, HW, Cc = 3, 49, 88
    z, t = torch.randn(N, HW, Cc), torch.randn(N, HW, Cc)
    sc, sh = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.3
    a = F.silu(z * sc + sh)
    zd, td, scd, shd = (x.contiguous().to(dev()) for x in (z, t, sc, sh))
    pool, dgate = torch.zeros(N, Cc, device=dev()), torch.zeros(N, Cc, device=dev())
    L.check(L.lib().mmvqa_se_pool(L.stream_ptr(), P(zd), P(scd), P(shd), P(pool), N, HW, Cc))
    L.check(L.lib().mmvqa_se_dgate(L.stream_ptr(), P(td), P(zd), P(scd), P(shd), P(dgate), N, HW, Cc))
    torch.cuda.synchronize()
    assert_close(pool, a.mean(1), TOL, "se_pool")
    assert_close(dgate, (t * a).sum(1), TOL, "se_dgate")
    # du = (t * gate + add / HW) * silu'(z*sc+sh); sums of du and du*xhat
    gate, add = torch.rand(N, Cc), torch.randn(N, Cc)
    mu, istd = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    u = (z * sc + sh).requires_grad_(True)
    F.silu(u).backward(t * gate[:, None, :] + add[:, None, :] / HW)
    out = torch.zeros(N * HW, Cc, device=dev())
    stat = torch.zeros(L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=dev())
    gd, ad, mud, isd = (x.contiguous().to(dev()) for x in (gate, add, mu, istd))
    L.check(L.lib().mmvqa_act_bwd_stats(L.stream_ptr(), P(td), P(gd), P(ad), P(zd), P(scd), P(shd), P(mud), P(isd), L.ACT_SILU,
                                        P(out), P(stat), N * HW, HW, Cc))
    torch.cuda.synchronize()
    assert_close(out.view(N, HW, Cc), u.grad, TOL, "act_bwd_stats")
    st = stat.sum(0).cpu()
    assert_close(st[:, 0], u.grad.sum((0, 1)).double(), 1e-4, "sum du")
    assert_close(st[:, 1], (u.grad * ((z - mu) * istd)).sum((0, 1)).double(), 1e-4, "sum du xhat")
    # block end: out = act(z*sc+sh) + skip   and   out = (z*sc+sh) + skip
    idn = torch.randn(N * HW, Cc)
    idd = idn.to(dev())
    o = torch.zeros(N * HW, Cc, device=dev())
    L.check(L.lib().mmvqa_bn_act_add(L.stream_ptr(), P(zd), P(scd), P(shd), L.ACT_SILU, P(idd), None, None, 0, 0, P(o), N * HW, Cc))
    torch.cuda.synchronize()
    assert_close(o, a.view(-1, Cc) + idn, TOL, "bn_act_add silu + skip")
    L.check(L.lib().mmvqa_bn_act_add(L.stream_ptr(), P(zd), P(scd), P(shd), L.ACT_NONE, None, None, None, 0, 0, P(o), N * HW, Cc))
    torch.cuda.synchronize()
    assert_close(o, (z * sc + sh).view(-1, Cc), TOL, "bn_act_add plain")


@pytest.mark.parametrize("B,mid,rd", [(16, 1824, 76), (3, 88, 22), (37, 640, 40), (64, 3072, 128), (2, 50, 13)])
def test_squeeze_excite_fc_layers(B, mid, rd):
    """the two fully connected layers of the squeeze-excite gate (timm SqueezeExcite: conv_reduce -> SiLU ->
    conv_expand -> sigmoid on the pooled [B, mid] tensor), forward and every gradient vs plain torch; full-size
    tf_efficientnetv2_m shapes, batches that are not a multiple of 16, channel counts that are not multiples of 4/64;
    weight and bias gradients ACCUMULATE into what is already there"""
    torch.manual_seed(B * 1000 + rd)
    pool = torch.randn(B, mid, requires_grad=True)
    Wr = (torch.randn(rd, mid) / mid ** 0.5).requires_grad_(True)
    br = (torch.randn(rd) * 0.1).requires_grad_(True)
    We = (torch.randn(mid, rd) / rd ** 0.5).requires_grad_(True)
    be = (torch.randn(mid) * 0.1).requires_grad_(True)
    rpre_ref = pool @ Wr.t() + br
    r_ref = F.silu(rpre_ref)
    gpre_ref = r_ref @ We.t() + be
    gate_ref = torch.sigmoid(gpre_ref)
    dgate = torch.randn(B, mid)
    gate_ref.backward(dgate)
    d = lambda x: x.detach().contiguous().to(dev())  # noqa: E731
    pool_d, Wr_d, br_d, We_d, be_d, dgate_d = d(pool), d(Wr), d(br), d(We), d(be), d(dgate)
    rpre, r = torch.zeros(B, rd, device=dev()), torch.zeros(B, rd, device=dev())
    gpre, gate = torch.zeros(B, mid, device=dev()), torch.zeros(B, mid, device=dev())
    L.check(L.lib().mmvqa_se_fc_fwd(L.stream_ptr(), P(pool_d), P(Wr_d), P(br_d), P(We_d), P(be_d), P(rpre), P(r), P(gpre),
                                    P(gate), B, mid, rd))
    torch.cuda.synchronize()
    assert_close(rpre, rpre_ref.detach(), TOL, "rpre")
    assert_close(r, r_ref.detach(), TOL, "r")
    assert_close(gpre, gpre_ref.detach(), TOL, "gpre")
    assert_close(gate, gate_ref.detach(), TOL, "gate")
    base = [torch.randn(mid, rd), torch.randn(mid), torch.randn(rd, mid), torch.randn(rd)]   # gradients already there
    dWe, dbe, dWr, dbr = (x.clone().to(dev()) for x in base)
    dpool = torch.full((B, mid), float("nan"), device=dev())
    n = L.lib().mmvqa_se_fc_bwd_scratch_floats(B, mid, rd)
    scratch = torch.full((n,), float("nan"), device=dev())
    L.check(L.lib().mmvqa_se_fc_bwd(L.stream_ptr(), P(dgate_d), P(gpre), P(r), P(rpre), P(pool_d), P(We_d), P(Wr_d), P(dWe),
                                    P(dbe), P(dWr), P(dbr), P(dpool), P(scratch), B, mid, rd))
    torch.cuda.synchronize()
    assert_close(dpool, pool.grad, TOL, "dpool")
    assert_close(dWe.cpu() - base[0], We.grad, TOL, "dWe")
    assert_close(dbe.cpu() - base[1], be.grad, TOL, "dbe")
    assert_close(dWr.cpu() - base[2], Wr.grad, TOL, "dWr")
    assert_close(dbr.cpu() - base[3], br.grad, TOL, "dbr")


@pytest.mark.parametrize("C,pro,act", [(24, False, "serf"), (64, True, "serf"), (64, True, "relu"), (24, True, "serf")])
def test_tap_thin_forward(C, pro, act):
    """stem tap with few channels (conv1x1 C -> hidden, activation, global average pool; image_encoding.py:53-62) through
    the register-resident kernel: vs plain torch, images whose pixel count is a multiple of 32 but not of the wave's run
    of row tiles (image changes inside a wave), hidden sizes 768 and 100 (ragged last column group)"""
    torch.manual_seed(C + (7 if pro else 0))
    B, HW = 3, 12544
    M = B * HW
    for N in (768, 100):
        assert L.lib().mmvqa_tap_thin_ok(M, N, C, HW) == 1
        x = torch.randn(M, C)
        W = torch.randn(N, C) / C ** 0.5
        sc, sh = torch.rand(C) + 0.5, torch.randn(C) * 0.3
        xin = torch.relu(x * sc + sh) if pro else x
        u = xin @ W.t()
        ref = (O.serf(u) if act == "serf" else torch.relu(u)).view(B, HW, N).mean(1)
        xd, Wd, scd, shd = (t.contiguous().to(dev()) for t in (x, W, sc, sh))
        out = torch.zeros(B, N, device=dev())
        L.check(L.lib().mmvqa_tap_thin_fwd(L.stream_ptr(), P(xd), P(scd) if pro else None, P(shd) if pro else None, P(Wd),
                                           P(out), M, N, C, HW, ACT(act)))
        torch.cuda.synchronize()
        assert_close(out, ref, TOL, f"tap_thin N={N}")
        # backward recompute: du = dv[img] / HW * act'(u)
        dv = torch.randn(B, N)
        uu = u.clone().requires_grad_(True)
        (O.serf(uu) if act == "serf" else torch.relu(uu)).view(B, HW, N).mean(1).backward(dv)
        du = torch.full((M, N), float("nan"), device=dev())
        dvd = dv.to(dev())
        L.check(L.lib().mmvqa_tap_thin_bwd(L.stream_ptr(), P(xd), P(scd) if pro else None, P(shd) if pro else None, P(Wd),
                                           P(dvd), P(du), M, N, C, HW, ACT(act)))
        torch.cuda.synchronize()
        assert_close(du, uu.grad, TOL, f"tap_thin du N={N}")
    assert L.lib().mmvqa_tap_thin_ok(16 * 784, 768, 64, 784) == 0     # 784 pixels per image: not a multiple of 32
    assert L.lib().mmvqa_tap_thin_ok(M, 768, 256, HW) == 0             # deep contraction: the GEMM kernel's job

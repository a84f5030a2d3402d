"""Debug aid: 3x3 weight gradient through the general loaders vs the uniform-tap loaders (pixel table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ctypes as C
from hip_helpers import *

torch.manual_seed(0)
N, H, W, Cin, Cout, K, s, p = 2, 14, 14, 64, 64, 3, 1, 1
G = torch.randn(N * H * W, Cout, device=dev())
z = torch.randn(N * H * W, Cout, device=dev())
x = torch.randn(N * H * W, Cin, device=dev())
coef = [torch.rand(Cout, device=dev()) + 0.5, torch.randn(Cout, device=dev()) * 0.1, torch.randn(Cout, device=dev()) * 0.1]
sc, sh = torch.rand(Cin, device=dev()) + 0.5, torch.randn(Cin, device=dev()) * 0.1
tab = torch.zeros(N * H * W, dtype=torch.int32, device=dev())
L.check(L.lib().mmvqa_pixmask(L.stream_ptr(), P(tab), N, H, W, H, W, K, K, s, p))
for mode in ("plain", "dz", "relu", "both"):
    outs = []
    for use_tab in (False, True):
        out = torch.zeros(Cout, K * K * Cin, device=dev())
        d = conv_desc_wgrad(G, x, N, H, W, Cin, Cout, K, s, p, out)
        if mode in ("dz", "both"):
            d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(coef[0]), P(coef[1]), P(coef[2])
        if mode in ("relu", "both"):
            d.b_pro, d.b_c0, d.b_c1 = L.PRO_AFFINE_RELU, P(sc), P(sh)
        if use_tab:
            d.pixmask = P(tab)
        run_igemm(d, L.KIND_WGRAD, tile=3)
        outs.append(out.cpu())
    a, b = outs
    err = (a - b).abs()
    print(mode, "max err", float(err.max()), "ref max", float(a.abs().max()))
    if err.max() > 1e-3:
        e4 = err.view(Cout, K * K, Cin)
        print("  per-tap max err", [round(float(e4[:, t].max()), 3) for t in range(K * K)])
        print("  per-co-block max err", [round(float(e4[c:c + 16].max()), 3) for c in range(0, Cout, 16)])
        print("  ref[0,:3,0:3]", a.view(Cout, 9, Cin)[0, :3, :3].tolist(), "got", b.view(Cout, 9, Cin)[0, :3, :3].tolist())

import os, sys, math, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hip_helpers import *
import ctypes as C

def run(kind_name, general, tile):
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
{'os.environ["MMVQA_IGEMM_GENERAL"]="1"' if general else ''}
import torch, ctypes as C
from hip_helpers import *
torch.manual_seed(0)
N,H,W,Cin,Cout,K,s,p = 2,14,14,64,64,3,1,1
G = torch.randn(N*H*W, Cout, device=dev()); z = torch.randn(N*H*W, Cout, device=dev())
w = torch.randn(Cout, K*K*Cin, device=dev())*0.05
x = torch.randn(N*H*W, Cin, device=dev())
coef = [torch.rand(Cout, device=dev())+0.5, torch.randn(Cout, device=dev())*0.1, torch.randn(Cout, device=dev())*0.1]
variant = {kind_name!r}
if variant == 'dgrad_c0': coef[1].zero_(); coef[2].zero_()
if variant == 'dgrad_c01': coef[2].zero_()
if variant == 'dgrad_c02': coef[1].zero_()
if variant == 'dgrad_const': coef[0].fill_(1.0); coef[1].zero_(); coef[2].zero_()
sc, sh = torch.rand(Cin, device=dev())+0.5, torch.randn(Cin, device=dev())*0.1
if {kind_name!r}.startswith('dgrad_') or {kind_name!r} == 'dgrad':
    out = torch.zeros(N*H*W, Cin, device=dev())
    d = conv_desc_dgrad(G, w, N,H,W,Cin,Cout,K,s,p,out)
    d.A2, d.a_pro, d.a_c0, d.a_c1, d.a_c2 = P(z), L.PRO_DZ, P(coef[0]), P(coef[1]), P(coef[2])
    kd = L.KIND_DGRAD
elif {kind_name!r} == 'dgrad0':
    out = torch.zeros(N*H*W, Cin, device=dev())
    d = conv_desc_dgrad(G, w, N,H,W,Cin,Cout,K,s,p,out)
    kd = L.KIND_DGRAD
else:
    out = torch.zeros(Cout, K*K*Cin, device=dev())
    d = conv_desc_wgrad(G, x, N,H,W,Cin,Cout,K,s,p,out)
    kd = L.KIND_WGRAD
L.check(L.lib().mmvqa_igemm(C.byref(d), kd, 0, {tile}, L.stream_ptr()))
torch.cuda.synchronize()
torch.save(out.cpu(), '/tmp/dbg_{kind_name}_{int(general)}_{tile}.pt')
"""
    subprocess.check_call([sys.executable, "-c", code])

for kind in ("dgrad_const", "dgrad_c0", "dgrad_c01", "dgrad_c02"):
    for tile in (3,):
        run(kind, True, tile); run(kind, False, tile)
        a = torch.load(f"/tmp/dbg_{kind}_1_{tile}.pt"); b = torch.load(f"/tmp/dbg_{kind}_0_{tile}.pt")
        err = (a - b).abs()
        print(kind, "tile", tile, "max err", float(err.max()), "ref max", float(a.abs().max()))
        if err.max() > 1e-3:
            bad = (err > 1e-3)
            rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
            print("  bad rows", rows.numel(), rows[:20].tolist(), "bad cols", cols.numel(), cols[:20].tolist())
            r = int(rows[0]); print("  row", r, "(n,y,x)=", r // 196, (r % 196) // 14, r % 14, "ref", a[r, :4].tolist(), "got", b[r, :4].tolist())

"""Randomised shape stress of the bf16x3 conv / weight-gradient kernels against torch (fp64 reference)."""
import sys, os, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
import torch.nn.functional as F
dev = "cuda"
random.seed(1); torch.manual_seed(1)
worst = 0.0
for it in range(60):
    N = random.choice([1, 1, 2, 3]); cin = random.choice([1, 3, 8, 16, 17, 32, 40, 64, 96]); cout = random.choice([1, 2, 27, 32, 33, 64, 100])
    D = random.randint(1, 9); H = random.randint(1, 20); W = random.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 20, 31, 32, 33, 48, 60])
    x = torch.randn(N, cin, D, H, W, device=dev, requires_grad=True); w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1).requires_grad_()
    ops.CONV_X3 = True
    y = ops._Conv3d.apply(x, None, w, 1, False)
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yr = F.conv3d(xd, wd, None, 1, 1)
    gxr, gwr = torch.autograd.grad((yr * gy.double()).sum(), [xd, wd])
    def rel(a, b): return ((a.double() - b).abs().max() / (b.abs().max() + 1e-30)).item()
    e = max(rel(y, yr), rel(gx, gxr), rel(gw, gwr))
    worst = max(worst, e)
    flag = "" if e < 2e-5 else "   <-- CHECK"
    print(f"{it:2d} N{N} {cin:3d}->{cout:3d} {D}x{H}x{W:2d}: fwd {rel(y, yr):.1e} dx {rel(gx, gxr):.1e} dw {rel(gw, gwr):.1e}{flag}")
    assert torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(gw).all()
print("worst relative error", worst)
assert worst < 2e-5

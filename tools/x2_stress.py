"""Randomised stress of the f16x2 conv / weight-gradient kernels against fp64: shapes on and off the tile / alignment grids,
operand magnitudes over 12 decades, training form (autograd: forward, backward-data, weight gradient) and the fused
inference epilogue (affine + LeakyReLU + two residual streams).  python tools/x2_stress.py [iterations]"""
import sys, os, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
import torch.nn.functional as F
dev = "cuda"
random.seed(2); torch.manual_seed(2)
assert ops.CONV_X2 and ops.CONV_X3
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
worst = 0.0


def rel(a, b):
    return ((a.double() - b).abs().max() / (b.abs().max() + 1e-300)).item()


for it in range(iters):
    big = it % 10 == 9                       # every tenth case: many tiles per workgroup, several samples
    N = random.choice([2, 3, 5]) if big else random.choice([1, 1, 2, 3])
    cin = random.choice([1, 3, 8, 16, 17, 32, 40, 64, 96]); cout = random.choice([1, 2, 27, 32, 33, 64, 100])
    if big:
        cin, cout = random.choice([16, 32, 40]), random.choice([32, 64])
    D = random.randint(8, 20) if big else random.randint(1, 9)
    H = random.randint(20, 50) if big else random.randint(1, 20)
    W = random.choice([48, 64, 100, 120]) if big else random.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 20, 31, 32, 33, 48, 60])
    mx, mg = 10.0 ** random.uniform(-8, 4), 10.0 ** random.uniform(-10, 2)
    x = (torch.randn(N, cin, D, H, W, device=dev) * mx).requires_grad_()
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 10.0 ** random.uniform(-3, 0)).requires_grad_()
    y = ops._Conv3d.apply(x, None, w, 1, False)
    gy = torch.randn_like(y) * mg
    gx, gw = torch.autograd.grad((y * gy).sum(), [x, w])
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yr = F.conv3d(xd, wd, None, 1, 1)
    gxr, gwr = torch.autograd.grad((yr * gy.double()).sum(), [xd, wd])
    # fused inference epilogue
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * mx
    rp = torch.randn_like(y) * mx; rq = torch.randn_like(y) * mx
    with torch.no_grad():
        ye = ops._conv_sliced(x.detach(), None, w.detach(), cin, cout, 27, 0, 0, 3, 1, False, sc, sh, 0.1, rp, rq, emit_amax=True)
    yer = F.leaky_relu(yr.detach() * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1) + rp.double(), 0.1) + rq.double()
    word = ye._dca_amax[0].view(torch.float32).max().item()
    e = max(rel(y, yr), rel(gx, gxr), rel(gw, gwr), rel(ye, yer))
    worst = max(worst, e)
    ok = all(torch.isfinite(t).all() for t in (y, gx, gw, ye)) and word == ye.abs().max().item()
    flag = "" if (e < 1e-5 and ok) else "   <-- CHECK"
    if flag or it % 10 == 9:
        print(f"{it:3d} N{N} {cin:3d}->{cout:3d} {D}x{H}x{W:3d} |x|~{mx:.0e} |gy|~{mg:.0e}: fwd {rel(y, yr):.1e} dx {rel(gx, gxr):.1e} "
              f"dw {rel(gw, gwr):.1e} epilogue {rel(ye, yer):.1e}{flag}", flush=True)
    assert ok
# weight gradient of the stride-2 / transposed convolution (conv3d_wgrad_s2_f16x2.hip): fine W % 8 == 0
for it in range(iters // 3):
    N = random.choice([1, 2, 3]); cx = random.choice([3, 16, 20, 32, 40, 64]); cy = random.choice([8, 32, 33, 64, 100])
    D, H, W = random.randint(1, 9), random.randint(1, 22), random.choice([8, 16, 24, 40, 64, 72, 136])
    cd = tuple((v + 1) // 2 for v in (D, H, W))
    x = torch.randn(N, cx, D, H, W, device=dev) * 10.0 ** random.uniform(-6, 3)
    dy = torch.randn(N, cy, *cd, device=dev) * 10.0 ** random.uniform(-10, 1)
    gw = torch.empty(cy, cx, 3, 3, 3, device=dev)
    before = ops.AMAX_STATS["computed"]
    ops._wgrad(x, dy, gw, 0, cx, cy, 3, 2, cx * 27, 27)
    assert ops.AMAX_STATS["computed"] == before + 2, "the f16x2 stride-2 kernel was not taken"
    ref = torch.nn.grad.conv3d_weight(x.double(), (cy, cx, 3, 3, 3), dy.double(), stride=2, padding=1)
    e = rel(gw, ref)
    worst = max(worst, e)
    if e >= 1e-5 or not torch.isfinite(gw).all() or it % 10 == 9:
        print(f"s2 {it:3d} N{N} {cx:3d}<->{cy:3d} fine {D}x{H}x{W:3d}: dw {e:.1e}" + ("" if e < 1e-5 else "   <-- CHECK"), flush=True)
    assert torch.isfinite(gw).all()
print("worst relative error", worst, "| maxima:", ops.AMAX_STATS)
assert worst < 1e-5

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"
torch.manual_seed(0)
def wg(x, dy, x3):
    ops.CONV_X3 = x3
    Cx, Cy = x.shape[1], dy.shape[1]
    gw = torch.empty(Cy, Cx, 3, 3, 3, device=dev)
    ops._wgrad(x, dy, gw, 0, Cx, Cy, 3, 1, Cx * 27, 27)
    return gw
for (N, Cx, Cy, D, H, W) in [(1, 32, 32, 2, 4, 16), (1, 32, 32, 4, 8, 32), (2, 40, 32, 5, 7, 20), (1, 64, 33, 3, 9, 36), (1, 16, 27, 6, 6, 12)]:
    x = torch.randn(N, Cx, D, H, W, device=dev); dy = torch.randn(N, Cy, D, H, W, device=dev)
    ref = torch.nn.grad.conv3d_weight(x.double(), (Cy, Cx, 3, 3, 3), dy.double(), padding=1)
    g3 = wg(x, dy, True); g32 = wg(x, dy, False)
    sc = ref.abs().max().item()
    print((N, Cx, Cy, D, H, W), "x3 err %.3e  fp32 err %.3e  (scale %.1f)" % ((g3.double() - ref).abs().max().item(), (g32.double() - ref).abs().max().item(), sc))
x = torch.randn(1, 32, 48, 136, 240, device=dev).relu_(); dy = torch.randn(1, 32, 48, 136, 240, device=dev)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("wgrad 32x32 @48x136x240: x3 %.3f ms   fp32 %.3f ms" % (t(lambda: wg(x, dy, True)), t(lambda: wg(x, dy, False))))
g3 = wg(x, dy, True); g32 = wg(x, dy, False)
print("full-size x3 vs fp32 rel-l2 %.3e" % ((g3 - g32).norm() / g32.norm()).item())
g3b = wg(x, dy, True); print("bitwise reproducible:", torch.equal(g3, g3b))

"""Weight gradient of the stride-2 convolution 32 -> 64 (fine 48x136x240, coarse 24x68x120): f16x2 split kernel
(conv3d_wgrad_s2_f16x2.hip) against the fp32 MFMA kernel, error vs fp64 on a small case and us per call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def wg(x, dy, on):
    ops.WGRAD_S2_X2 = on
    cx, cy = x.shape[1], dy.shape[1]
    gw = torch.empty(cy, cx, 3, 3, 3, device=dev)
    ops._wgrad(x, dy, gw, 0, cx, cy, 3, 2, cx * 27, 27)
    return gw


torch.manual_seed(0)
x = torch.randn(2, 32, 6, 10, 40); dy = torch.randn(2, 64, 3, 5, 20)
ref = torch.nn.grad.conv3d_weight(x.double(), (64, 32, 3, 3, 3), dy.double(), stride=2, padding=1)
for on in (True, False):
    g = wg(x.to(dev), dy.to(dev), on)
    print("f16x2" if on else "fp32 ", "max err vs fp64 rel. to max: %.2e" % ((g.cpu().double() - ref).abs().max() / ref.abs().max()).item())
for N in (1, 4):
    x = torch.randn(N, 32, 48, 136, 240, device=dev).relu_(); dy = torch.randn(N, 64, 24, 68, 120, device=dev)
    ops._exps_of(x); ops._exps_of(dy)
    print("N=%d 32->64 s2: f16x2 %.1f us | fp32 MFMA %.1f us" % (N, t(lambda: wg(x, dy, True)), t(lambda: wg(x, dy, False))), flush=True)

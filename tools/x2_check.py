"""f16x2 split kernels (conv3d_f16x2.hip, conv3d_wgrad_f16x2.hip) next to the bf16x3 and fp32-MFMA families: error against
fp64 at small shapes (normal, tiny-magnitude and many-binade operands), then forward / weight-gradient launch times at the
network's shapes.  python tools/x2_check.py [--notime]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
import torch.nn.functional as F
dev = "cuda"


def fam(name):
    ops.CONV_X3 = name != "fp32"
    ops.CONV_X2 = name == "x2"


def conv(x, w):
    N, Cin = x.shape[:2]
    return ops._conv_sliced(x, None, w, Cin, w.shape[0], 27, 0, 0, 3, 1, False)


def wgrad(x, dy):
    cx, cy = x.shape[1], dy.shape[1]
    gw = torch.empty(cy, cx, 3, 3, 3, device=dev)
    ops._wgrad(x, dy, gw, 0, cx, cy, 3, 1, cx * 27, 27)
    return gw


torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
cases = []
for (N, Cin, Cout, D, H, W) in [(1, 32, 32, 8, 16, 32), (2, 40, 64, 7, 13, 21), (1, 64, 33, 5, 9, 17)]:
    cases.append(("randn", torch.randn(N, Cin, D, H, W, generator=g), torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.05))
x = torch.randn(1, 32, 6, 12, 32, generator=g)
cases.append(("relu", x.relu(), torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05))
cases.append(("x*1e-9", x * 1e-9, torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05))
cases.append(("x*1e6,w*1e-4", x * 1e6, torch.randn(32, 32, 3, 3, 3, generator=g) * 1e-4))
cases.append(("binades", x * torch.exp2(torch.randint(-12, 12, x.shape, generator=g).float()),
              torch.randn(32, 32, 3, 3, 3, generator=g) * torch.exp2(torch.randint(-6, 6, (32, 32, 3, 3, 3), generator=g).float())))
cases.append(("zeros", x * 0, torch.randn(32, 32, 3, 3, 3, generator=g)))
for name, x, w in cases:
    ref = F.conv3d(x.double(), w.double(), padding=1)
    sc = ref.abs().max().item() + 1e-300
    dy = torch.randn(ref.shape, generator=g) * (1e-7 if "1e-9" in name else 1.0)
    refw = torch.nn.grad.conv3d_weight(x.double(), w.shape, dy.double(), padding=1)
    scw = refw.abs().max().item() + 1e-300
    out = []
    for f in ("x2", "x3", "fp32"):
        fam(f)
        y = conv(x.to(dev), w.to(dev))
        gw = wgrad(x.to(dev), dy.to(dev))
        out.append("%s fwd %.2e dw %.2e" % (f, (y.cpu().double() - ref).abs().max().item() / sc,
                                             (gw.cpu().double() - refw).abs().max().item() / scw))
    print("%-14s %s | rel. to max" % (name, " | ".join(out)), flush=True)
print("amax words:", ops.AMAX_STATS)
if "--notime" in sys.argv:
    sys.exit(0)


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


for (N, cx, cy, d, h, w) in [(1, 32, 32, 48, 136, 240), (4, 32, 32, 48, 136, 240), (1, 64, 64, 24, 68, 120), (1, 40, 32, 48, 136, 240)]:
    x = torch.randn(N, cx, d, h, w, device=dev).relu_(); dy = torch.randn(N, cy, d, h, w, device=dev)
    wt = torch.randn(cy, cx, 3, 3, 3, device=dev) * 0.05
    line = []
    for f in ("x2", "x3"):
        fam(f)
        with ops.frozen_weights():
            if f == "x2":
                ops._amax_of(x); ops._amax_of(dy)      # producers emit these in the network
            line.append("%s fwd %.1f us wgrad %.1f us" % (f, t(lambda: conv(x, wt)), t(lambda: wgrad(x, dy))))
    print("N=%d %d->%d @%dx%dx%d: %s" % (N, cx, cy, d, h, w, " | ".join(line)), flush=True)
x = torch.randn(4, 32, 48, 136, 240, device=dev)
word = torch.empty(ops.AMAX_SLOTS, dtype=torch.int32, device=dev)
print("dca_amax_f32 over 4x32x48x136x240: %.1f us" % t(lambda: ops._chk(ops._L().dca_amax_f32(ops._ptr(x), x.numel(), ops._ptr(word), ops._stream()), "amax")))

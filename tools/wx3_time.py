"""Weight gradient of the 3x3x3 stride-1 convolution (bf16x3 kernel + slab reduce) at the network's shapes, us per call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, cx, cy, d, h, w) in [(1, 32, 32, 48, 136, 240), (4, 32, 32, 48, 136, 240), (1, 64, 64, 24, 68, 120)]:
    x = torch.randn(N, cx, d, h, w, device=dev).relu_(); dy = torch.randn(N, cy, d, h, w, device=dev)
    gw = torch.empty(cy, cx, 3, 3, 3, device=dev)
    print("N=%d %d->%d @%dx%dx%d: %.1f us" % (N, cx, cy, d, h, w, t(lambda: ops._wgrad(x, dy, gw, 0, cx, cy, 3, 1, cx * 27, 27))))

"""The 32 -> 1 logit head at the batch-4 layer shape: forward, backward-data, weight gradient (HIP events)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
x = torch.randn(4, 32, 48, 136, 240, device=dev); w = (torch.randn(1, 32, 3, 3, 3, device=dev) * 0.05)
gy = torch.randn(4, 1, 48, 136, 240, device=dev)


def timed(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    print(f"forward            {timed(lambda: ops.conv3d(x, w, 1, False)):8.1f} us")
for fused in (True, False):
    ops.C1_WGRAD_FUSED = fused
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = ops.conv3d(xg, wg, 1, False)
    print(f"weight gradient ({'fused' if fused else 'expand'}) {timed(lambda: torch.autograd.grad(y, [wg], gy, retain_graph=True)):8.1f} us")
    if fused:
        print(f"backward-data      {timed(lambda: torch.autograd.grad(y, [xg], gy, retain_graph=True)):8.1f} us")

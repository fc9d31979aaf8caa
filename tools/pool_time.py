"""avgpool3d backward (with the fused second gradient) at the batch-4 layer shape: launch time via HIP events"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
lib = ops._L()
gy = torch.randn(4, 32, 24, 68, 120, device=dev); res = torch.randn(4, 32, 48, 136, 240, device=dev); gx = torch.empty_like(res)
f = lambda: ops._chk(lib.dca_avgpool3d_bwd(ops._ptr(gy), ops._ptr(gx), ops._ptr(res), None, 128, 48, 136, 240, ops._stream()), "pool bwd")
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"avgpool3d_bwd + res, 4x32x48x136x240: {t * 1e3:.1f} us = {(gy.numel() + 2 * res.numel()) * 4 / t / 1e9:.2f} TB/s")
ref = torch.nn.functional.avg_pool3d(res.clone().requires_grad_(), 3, 2, 1)
x = res.clone().requires_grad_(); torch.nn.functional.avg_pool3d(x, 3, 2, 1).backward(gy)
print("max err vs torch", (gx - (x.grad + res)).abs().max().item())

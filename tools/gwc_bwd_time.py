"""gwc volume backward at the batch-4 shape (4 x 320 x 136 x 240, 40 groups, 48 disparities): launch time (HIP events)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
lib = ops._L()
B, C, H, W, D, G = 4, 320, 136, 240, 48, 40
L, R = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
gv = torch.randn(B, G, D, H, W, device=dev); gL, gR = torch.empty_like(L), torch.empty_like(R)
f = lambda: ops._chk(lib.dca_gwc_volume_bwd(ops._ptr(gv), ops._ptr(L), ops._ptr(R), ops._ptr(gL), ops._ptr(gR), B, C, H, W, D, G, ops._stream()), "gwc bwd")
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"gwc_bwd: {t * 1e3:.1f} us = {(gv.numel() + 4 * L.numel()) * 4 / t / 1e9:.2f} TB/s algorithmic")

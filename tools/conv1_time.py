import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
x = torch.randn(1, 32, 48, 136, 240, device=dev); x2 = torch.randn_like(x)
for cout, two in ((32, False), (27, False), (32, True)):
    cin = 64 if two else 32
    w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.1
    us = t(lambda: ops._conv_sliced(x, x2 if two else None, w, cin, cout, 1, 0, 0, 1, 1, False))
    mb = (cin + cout) * x[0, 0].numel() * 4 / 1e6
    print("conv1 %d->%d%s: %.1f us  (%.0f MB -> %.2f TB/s)" % (cin, cout, " (two inputs)" if two else "", us, mb, mb / us))

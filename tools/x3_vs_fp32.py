"""bf16x3 vs fp32-MFMA 3x3x3 stride-1 conv on the shapes the networks use (ms per launch)."""
import os, sys, torch
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
    return e0.elapsed_time(e1) / reps
for (cin, cout, d, h, w) in [(32, 32, 48, 136, 240), (40, 32, 48, 136, 240), (64, 32, 48, 136, 240), (64, 64, 24, 68, 120), (128, 128, 12, 34, 60), (32, 32, 24, 68, 120), (32, 32, 48, 96, 312), (32, 32, 16, 64, 128)]:
    x = torch.randn(1, cin, d, h, w, device=dev).relu_(); wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    res = {}
    for mode in (True, False):
        ops.CONV_X3 = mode; ops._X3_MIN_WORKGROUPS = 1
        res[mode] = t(lambda: ops._conv_sliced(x, None, wt, cin, cout, 27, 0, 0, 3, 1, False))
    gf = 2 * 27 * cin * cout * d * h * w / 1e9
    print("%3d->%3d @%dx%dx%d: x3 %.3f ms (%.0f TF/s)  fp32 %.3f ms (%.0f TF/s)  ratio %.2f" % (cin, cout, d, h, w, res[True], gf / res[True], res[False], gf / res[False], res[False] / res[True]))

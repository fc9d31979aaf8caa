"""bf16x3 conv 32->32 @48x136x240 with and without the fused BatchNorm statistics, and the separate statistics pass."""
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
for N in (1, 4):
    x = torch.randn(N, 32, 48, 136, 240, device=dev).relu_(); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    shift = torch.zeros(32, device=dev)
    bn = torch.nn.BatchNorm3d(32).to(dev)
    plain = t(lambda: ops._conv_sliced(x, None, w, 32, 32, 27, 0, 0, 3, 1, False))
    fused = t(lambda: ops._Conv3d.apply(x, None, w, 1, False, True))
    y = ops._conv_sliced(x, None, w, 32, 32, 27, 0, 0, 3, 1, False)
    st = t(lambda: ops.bn_stats_vector(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, 0.1, 1e-5))
    print("N=%d: conv %.1f us, conv+stats fused %.1f us (+%.1f), separate statistics pass (stats + finalize) %.1f us" % (N, plain, fused, fused - plain, st))

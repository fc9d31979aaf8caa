"""Launch time of the stride-2 3x3x3 convolution 32 -> 64 at the fine shape 48x136x240 (cost_agg.conv1 forward / the backward-data
of cost_agg.conv3): f16x2 kernel (conv3d_s2_f16x2.hip) against the fp32 MFMA kernel, batch 1 and 4; error of both against fp64 on
a small shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05


def timed(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for N in (1, 4):
    x = torch.relu(torch.randn(N, 32, 48, 136, 240, device=dev))
    ops._exps_of(x)            # in the network the exponents come with the tensor
    for fam in ("x2", "fp32"):
        ops.CONV_S2_X2 = fam == "x2"
        t = timed(lambda: ops._conv_sliced(x, None, w, 32, 64, 27, 0, 0, 3, 2, False))
        fl = 2.0 * 27 * 32 * 64 * N * 24 * 68 * 120
        print(f"N={N} {fam:5s}: {t:.4f} ms  = {fl / t / 1e9:.1f} TFLOP/s algorithmic", flush=True)
ops.CONV_S2_X2 = True

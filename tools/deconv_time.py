"""Transposed 3x3x3 convolution 64 -> 32 at 24x68x120 -> 48x136x240 (cost_agg.conv3 / backward-data of cost_agg.conv1):
bf16x3 kernel (deconv3d_x3.hip) against the fp32 MFMA kernel, plain and with the fused inference epilogue."""
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
for N in (1, 4):
    x = torch.randn(N, 64, 24, 68, 120, device=dev)
    w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
    sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev)
    rp, rq = torch.randn(N, 32, 48, 136, 240, device=dev), torch.randn(N, 32, 48, 136, 240, device=dev)
    flop = 2.0 * 64 * 32 * 27 * x[0, 0].numel() * N
    for x3 in (True, False):
        ops.CONV_X3 = x3
        us = t(lambda: ops.conv3d(x, w, 2, True))
        usf = t(lambda: ops.conv3d_fused_inference(x, w, 2, True, sc, sh, 0.0, rp, rq))
        print("N=%d %s: plain %.1f us (%.1f TFLOP/s fp32-equivalent), fused epilogue %.1f us" % (
            N, "bf16x3" if x3 else "fp32  ", us, flop / us / 1e6, usf))
    y3 = None
    ops.CONV_X3 = True; y3 = ops.conv3d(x, w, 2, True)
    ops.CONV_X3 = False; y32 = ops.conv3d(x, w, 2, True)
    print("   max |x3 - fp32| = %.3g (max |y| %.3g)" % ((y3 - y32).abs().max().item(), y32.abs().max().item()))

"""Micro-benchmark of the reduced-precision 3x3x3 conv (conv3d_lp.hip) at the BASELINE volume size, next to the
fp32-grade bf16x3 kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dcanet_amd
from dcanet_amd import ops

d, h, w = 48, 136, 240
dev = "cuda"
wgt = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad(), ops.frozen_weights():
    x32 = torch.randn(1, 32, d, h, w, device=dev)
    if not os.environ.get("DCA_LP_ONLY"):
        ms = timeit(lambda: ops.conv3d_fused_inference(x32, wgt, 1, False, sc, sh, 0.0))
        print(f"bf16x3 fp32->fp32            {ms*1e3:8.1f} us   {401.2/ms/1e3:6.2f} TB/s algorithmic")
    for lp in (torch.bfloat16, torch.float16):
        xl = x32.to(lp)
        for xin, odt, name, mb in ((x32, torch.float32, "f32->f32", 401.2), (x32, lp, "f32->lp ", 300.9),
                                   (xl, lp, "lp ->lp ", 200.6)):
            ms = timeit(lambda: ops.conv3d_lp(xin, wgt, lp, sc, sh, 0.0, None, None, odt))
            print(f"lp {str(lp)[6:]:9s} {name}        {ms*1e3:8.1f} us   {mb/ms/1e3:6.2f} TB/s algorithmic")

with torch.no_grad(), ops.frozen_weights():
    w1 = torch.randn(32, 32, 1, 1, 1, device=dev) * 0.1
    w2 = torch.randn(32, 64, 1, 1, 1, device=dev) * 0.1
    for lp in (torch.bfloat16, torch.float16):
        xl = x32.to(lp)
        ms = timeit(lambda: ops.conv1x1_lp(xl, w1, lp, None, sc, sh, 0.1))
        print(f"conv1 lp {str(lp)[6:]:9s} 32->32       {ms*1e3:8.1f} us   {200.6/ms/1e3:6.2f} TB/s algorithmic")
        ms = timeit(lambda: ops.conv1x1_lp(xl, w2, lp, xl, sc, sh, 1.0))
        print(f"conv1 lp {str(lp)[6:]:9s} 64->32 (2 in) {ms*1e3:8.1f} us   {300.9/ms/1e3:6.2f} TB/s algorithmic")
    ms = timeit(lambda: ops.conv3d_fused_inference(x32, w1, 1, False, sc, sh, 0.1))
    print(f"conv1 fp32 32->32                {ms*1e3:8.1f} us   {401.2/ms/1e3:6.2f} TB/s algorithmic")
    ms = timeit(lambda: ops.conv3d_fused_inference(x32, w2, 1, False, sc, sh, 1.0, x2=x32))
    print(f"conv1 fp32 64->32 (2 in)         {ms*1e3:8.1f} us   {601.8/ms/1e3:6.2f} TB/s algorithmic")

with torch.no_grad(), ops.frozen_weights():
    xc = torch.randn(1, 64, d // 2, h // 2, w // 2, device=dev)
    wd = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
    for lp in (torch.bfloat16, torch.float16):
        rp_ = torch.randn(1, 32, d, h, w, device=dev).to(lp)
        for exact in (True, False):
            ms = timeit(lambda: ops.deconv3d_lp(xc, wd, lp, sc, sh, 0.0, rp_, None, exact=exact))
            print(f"deconv 64->32 {str(lp)[6:]:9s} {'fp32 MFMA, 2-byte out' if exact else 'lp kernel            '} {ms*1e3:8.1f} us   "
                  f"{(50.1 + 200.6) / ms / 1e3:6.2f} TB/s algorithmic")

with torch.no_grad(), ops.frozen_weights():
    ws2 = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
    sc64, sh64 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    for lp in (torch.bfloat16, torch.float16):
        xl = x32.to(lp)
        for exact in (True, False):
            ms = timeit(lambda: ops.conv3d_s2_lp(xl, ws2, sc64, sh64, 0.0, exact=exact))
            print(f"conv s2 32->64 {str(lp)[6:]:9s} {'fp32 MFMA, 2-byte in ' if exact else 'lp kernel            '} {ms*1e3:8.1f} us   "
                  f"{(100.3 + 50.1) / ms / 1e3:6.2f} TB/s algorithmic")

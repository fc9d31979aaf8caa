"""In-kernel s_memtime stamps of conv3_f16x2_kernel (build with DCA_EXTRA_CFLAGS=-DX2_STAMP=1; DCA_CONV=x3 and -DX3_STAMP=1 for conv3_bf16x3_kernel): phase timeline of
workgroup 0, wave 0 on the 32->32 convolution at 48x136x240 (6 phases of 54 MFMAs per wave and tile)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"
x = torch.randn(1, 32, 48, 136, 240, device=dev).relu_()
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
stamps = torch.zeros(96 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    y = ops._conv_sliced(x, None, w, 32, 32, 27, 0, 0, 3, 1, False, res_post=stamps.view(torch.float32))
torch.cuda.synchronize()
s = stamps.cpu().view(96, 8)
t0 = int(s[0, 0])
for k in range(30):
    a = [int(v) - t0 for v in s[k]]
    line = "phase %2d start %7d | slabs %5d | mfma %5d |" % (k, a[0], a[1] - a[0], a[2] - a[1])
    if a[3] > 0:
        line += " bar %5d store_B %5d bar %5d |" % (a[3] - a[2], a[4] - a[3], a[5] - a[4])
    else:
        line += " bar %5d                         |" % (a[5] - a[2])
    if a[7] > 0:
        line += " epilogue %5d" % (a[7] - a[6])
    print(line)

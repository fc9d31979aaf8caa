"""In-kernel s_memtime stamps of wgrad3_f16x2_kernel (build with DCA_EXTRA_CFLAGS=-DWX2_STAMP=1): tile timeline of
workgroup 0, waves 0 (7 taps) and 3 (6 taps) on the 32->32 weight gradient at 48x136x240."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"
lib = ops._L()
x = torch.randn(1, 32, 48, 136, 240, device=dev).relu_()
dy = torch.randn(1, 32, 48, 136, 240, device=dev)
xa, ya = ops._amax_of(x), ops._amax_of(dy)
nws = lib.dca_conv3d_wgrad_x2_workspace(1, 32, 32, 48, 136, 240)
part = torch.zeros(nws + 2 * 2 * 64 * 8, device=dev, dtype=torch.float32)     # + the stamp buffer (int64 x 1024)
dw = torch.empty(32, 32, 3, 3, 3, device=dev)
for _ in range(3):
    ops._chk(lib.dca_conv3d_wgrad_x2(ops._ptr(x), ops._ptr(xa), ops._ptr(dy), ops._ptr(ya), ops._ptr(part), ops._ptr(dw), 1, 32, 32,
                                     48, 136, 240, 32 * 27, 27, ops._stream()), "wgrad_x2")
torch.cuda.synchronize()
s = part[nws:].view(torch.int64).cpu().view(2, 64, 8)
t0 = int(s[0, 0, 0])
for k in range(12):
    row = []
    for g in range(2):
        a = [int(v) - t0 for v in s[g, k]]
        row.append("w%d start %6d loads %4d mfma %5d bar %5d store %5d bar %4d" % (3 * g, a[0], a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4]))
    print("tile %2d | %s | %s" % (k, row[0], row[1]))

"""In-kernel s_memtime stamps of deconv3_bf16x3_kernel (build with DCA_EXTRA_CFLAGS=-DDX3_STAMP=1): slot timeline of
workgroup 0, waves 0 (half-group 0) and 4 (half-group 1)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(N, 64, 24, 68, 120, device=dev)
w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
stamps = torch.zeros(2 * 96 * 6, dtype=torch.int64, device=dev)
for _ in range(3):
    y = ops._conv_sliced(x, None, w, 64, 32, 27, 1, 0, 3, 2, True, res_post=stamps.view(torch.float32))
torch.cuda.synchronize()
s = stamps.cpu().view(2, 96, 6)
t0 = int(s[:, 0, 0].min())
for k in range(40):
    row = []
    for g in range(2):
        a = [int(v) - t0 for v in s[g, k]]
        role = "C" if (k + g) % 2 == 0 else "L"
        if role == "C":
            row.append("g%d C start %6d mfma %5d epi %5d            bar %5d" % (g, a[0], a[3] - a[0], a[2] - a[3], a[5] - a[2]))
        else:
            row.append("g%d L start %6d wait %5d st %5d ld %5d bar %5d" % (g, a[0], a[4] - a[0], a[1] - a[4], a[3] - a[1], a[5] - a[3]))
    print("slot %2d | %s | %s" % (k, row[0], row[1]))

"""In-kernel timeline of conv3_f16x2_kernel (X2_STAMP build): s_memtime stamps of workgroup 0, waves 0 and 7, 8 marks per
chunk: 0 chunk start, 1 staging loads issued + first fragments requested, 2 MFMAs issued, 3 staged data written to the other
image pair, 5 barrier passed, 6 / 7 (last chunk of a tile) epilogue start / end.

    bash tools/x2_stamp_run.sh        # builds tools/bin/libdca_stamp.so (-DX2_STAMP=1) here; run this script on the GPU box
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcanet_amd  # noqa: E402,F401
from dcanet_amd import ops  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libdca_stamp.so"))
p, i, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
lib.dca_conv3d_x2_forward.argtypes = [p, i, p, p, p, p, p, p, p, f, p] + [i] * 6 + [p]
lib.dca_conv3d_x2_forward_stats.argtypes = [p, i, p, p, p, p] + [i] * 6 + [p]
lib.dca_x2_debug_set_stamps.argtypes = [p]
dev = "cuda"
packed = "--fp32" not in sys.argv
stats = "--stats" in sys.argv
N, C, d, h, w = 1, 32, 48, 136, 240
x = torch.relu(torch.randn(N, C, d, h, w, device=dev))
wgt = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
L = ops._L()
xin = ops.pack_x2(x) if packed else x
xex = ops._exps_of(xin)
wx = torch.empty((L.dca_conv3d_x2_weight_bytes(C, C) // 2,), device=dev, dtype=torch.int16)
ops._chk(L.dca_conv3d_x2_prep_weight(ops._ptr(wgt), ops._ptr(wx), C, C, 0, 0, None, 0, ops._ptr(xex), ops._stream()), "prep")
y = torch.empty_like(x)
stamps = torch.zeros(2 * 96 * 8, dtype=torch.int64, device=dev)
part = torch.empty((C * 256 * 4,), device=dev, dtype=torch.float64)
for it in range(3):
    stamps.zero_()
    lib.dca_x2_debug_set_stamps(ops._ptr(stamps))
    if stats:
        rc = lib.dca_conv3d_x2_forward_stats(ops._ptr(xin), int(packed), ops._ptr(xex), ops._ptr(wx), ops._ptr(y), ops._ptr(part),
                                             N, C, C, d, h, w, ops._stream())
    else:
        rc = lib.dca_conv3d_x2_forward(ops._ptr(xin), int(packed), ops._ptr(xex), ops._ptr(wx), ops._ptr(y), None, None, None,
                                       None, 1.0, None, N, C, C, d, h, w, ops._stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
ref = torch.nn.functional.conv3d(x, wgt, padding=1)
print("packed" if packed else "fp32", "stats" if stats else "", "max err vs torch %.2e" % (y - ref).abs().max().item())
for slot, wv in enumerate((0, 7)):
    s = stamps[slot * 768:(slot + 1) * 768].view(96, 8).cpu()
    t0 = int(s[0, 0])
    print(f"wave {wv}: chunk rows (cycles since first mark; s_memtime ticks at 100 MHz x ... printed raw deltas)")
    for k in range(0, 16):
        row = [int(v) for v in s[k]]
        if row[0] == 0:
            break
        base = row[0]
        print("  k=%2d start %7d | loads+frag %5d  mfma %5d  stores %5d  barrier %5d | epi %s" % (
            k, row[0] - t0, row[1] - base, row[2] - row[1], row[3] - row[2], row[5] - row[3],
            ("%5d + %5d" % (row[6] - row[5], row[7] - row[6])) if row[6] else "-"))

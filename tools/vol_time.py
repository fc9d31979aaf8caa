"""Times the fused cost-volume builder at the BASELINE shape (fp32 / bf16 volume)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dcanet_amd
from dcanet_amd import ops
d, h, w = 48, 136, 240
fl, fr = torch.randn(1, 320, h, w, device="cuda"), torch.randn(1, 320, h, w, device="cuda")
segs = lambda t: (t[:, :64].contiguous(), t[:, 64:192].contiguous(), t[:, 192:].contiguous())
sl, sr = segs(fl), segs(fr)


def timeit(fn, reps=30):
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


with torch.no_grad():
    for dt, mb in ((torch.float32, 334.2), (torch.bfloat16, 208.9)):
        ms = timeit(lambda: ops.cost_volume(sl, sr, d, 40, out_dtype=dt))
        print(f"gwc_fused {str(dt)[6:]:9s} 3 segments {ms*1e3:7.1f} us  {mb/ms/1e3:5.2f} TB/s = {mb/ms/1e3/8:.3f} of 8 TB/s")

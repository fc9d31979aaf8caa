"""disparity attention forward + backward at the batch-4 shape of the cva blocks (4 x 32 x 24 x 68 x 120): kernel times"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
lib = ops._L()
B, C, n, HW = 4, 32, 24, 68 * 120
q, k, v, g = (torch.randn(B, C, n, HW, device=dev) for _ in range(4))
out, dq, dk, dv = (torch.empty_like(q) for _ in range(4))


def timed(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


P = ops._ptr
print(f"attn fwd {timed(lambda: ops._chk(lib.dca_disp_attention_fwd(P(q), P(k), P(v), P(out), B, C, n, HW, ops._stream()), 'fwd')):.1f} us")
print(f"attn bwd {timed(lambda: ops._chk(lib.dca_disp_attention_bwd(P(q), P(k), P(v), P(g), P(dq), P(dk), P(dv), B, C, n, HW, ops._stream()), 'bwd')):.1f} us")

"""BatchNorm kernels (stats, apply, backward) at 1/4 resolution, batch N: achieved HBM bandwidth."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
L = ops._L(); dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C, D, H, W = 32, 48, 136, 240
S = D * H * W
y = torch.randn(N, C, D, H, W, device=dev); dz = torch.randn_like(y); z = torch.empty_like(y); dy = torch.empty_like(y)
g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
stats = ops.bn_stats_vector(y, g, b, torch.zeros(C, device=dev), torch.ones(C, device=dev), True, 0.1, 1e-5)
nchunk = L.dca_bn_num_chunks(C, S)
part = torch.empty(C * nchunk * 2, device=dev, dtype=torch.float64); dgb = torch.empty(4 * C, device=dev)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
gb = N * C * S * 4 / 1e9
s = ops._stream
us = t(lambda: ops._chk(L.dca_bn_stats(ops._ptr(y), ops._ptr(part), N, C, S, s()), "stats"))
print("N=%d bn_stats    %.0f us  %.2f TB/s" % (N, us, gb / us * 1e3))
am = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=dev)
for word, tag in ((None, ""), (am, " + max |.| words")):
    us = t(lambda: ops._chk(L.dca_bn_apply(ops._ptr(y), ops._ptr(stats), None, None, ops._ptr(z), N, C, S, 0.0, ops._ptr(word), s()), "apply"))
    print("N=%d bn_apply%s    %.0f us  %.2f TB/s" % (N, tag, us, 2 * gb / us * 1e3))
    us = t(lambda: ops._chk(L.dca_bn_backward(ops._ptr(dz), ops._ptr(y), None, ops._ptr(stats), ops._ptr(part), ops._ptr(dgb), ops._ptr(dy), None, N, C, S, 0.0, 1, ops._ptr(word), s()), "bwd"))
    print("N=%d bn_backward%s %.0f us  %.2f TB/s (reduce 2 reads + apply 2 reads 1 write)" % (N, tag, us, 5 * gb / us * 1e3))
a = torch.empty_like(y)
us = t(lambda: torch.add(y, dz, out=a))
print("N=%d torch add   %.0f us  %.2f TB/s" % (N, us, 3 * gb / us * 1e3))
us = t(lambda: a.copy_(y))
print("N=%d torch copy  %.0f us  %.2f TB/s" % (N, us, 2 * gb / us * 1e3))

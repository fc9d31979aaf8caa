import os, sys, time, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
L = ops._L()
def x3(x, w, scale=None, shift=None, slope=1.0, res_pre=None, res_post=None, src_ab=0, flip=0):
    N, Cin, D, H, W = x.shape
    Cout = w.shape[1] if src_ab else w.shape[0]
    nb = L.dca_conv3d_x3_weight_bytes(Cin, Cout)
    wx = torch.empty(nb // 2, dtype=torch.int16, device=x.device)
    s = torch.cuda.current_stream().cuda_stream
    rc = L.dca_conv3d_x3_prep_weight(ops._ptr(w), ops._ptr(wx), Cin, Cout, src_ab, flip, s); assert rc == 0, rc
    y = torch.empty(N, Cout, D, H, W, device=x.device)
    rc = L.dca_conv3d_x3_forward(ops._ptr(x), ops._ptr(wx), ops._ptr(y), ops._ptr(scale), ops._ptr(shift), ops._ptr(res_pre), ops._ptr(res_post), float(slope), N, Cin, Cout, D, H, W, s); assert rc == 0, rc
    return y, wx
torch.manual_seed(0)
dev = "cuda"
for (N, Cin, Cout, D, H, W) in [(1, 32, 32, 8, 16, 32), (2, 40, 64, 7, 13, 21), (1, 16, 32, 4, 8, 16), (1, 64, 33, 5, 9, 17)]:
    x = torch.randn(N, Cin, D, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
    y, _ = x3(x, w)
    ref64 = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    ref32 = torch.nn.functional.conv3d(x, w, padding=1)
    y32 = ops._conv_sliced(x, None, w, Cin, Cout, 27, 0, 0, 3, 1, False)
    sc = ref64.abs().max().item()
    print((N, Cin, Cout, D, H, W), "x3 err vs f64 %.3e | fp32-mfma err %.3e | torch f32 err %.3e (scale %.2f)" % ((y.double()-ref64).abs().max().item(), (y32.double()-ref64).abs().max().item(), (ref32.double()-ref64).abs().max().item(), sc))
# epilogue + bwd-data form
N, Cin, Cout, D, H, W = 1, 32, 32, 6, 10, 20
x = torch.randn(N, Cin, D, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
scale = torch.rand(Cout, device=dev) + 0.5; shift = torch.randn(Cout, device=dev); rp = torch.randn(N, Cout, D, H, W, device=dev); rq = torch.randn_like(rp)
y, _ = x3(x, w, scale, shift, 0.1, rp, rq)
ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(x, w, padding=1) * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1) + rp, 0.1) + rq
print("epilogue err %.3e" % (y - ref).abs().max().item())
dy = torch.randn(N, Cout, D, H, W, device=dev)
gx, _ = x3(dy, w, src_ab=1, flip=1)
ref = torch.nn.grad.conv3d_input(x.shape, w, dy, padding=1)
print("bwd-data err %.3e" % (gx - ref).abs().max().item())
# timing at the benchmark shape
x = torch.randn(1, 32, 48, 136, 240, device=dev); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
y, wx = x3(x, w)
s = torch.cuda.current_stream().cuda_stream
def run():
    L.dca_conv3d_x3_forward(ops._ptr(x), ops._ptr(wx), ops._ptr(y), None, None, None, None, 1.0, 1, 32, 32, 48, 136, 240, s)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("x3 32->32 @48x136x240: %.3f ms  (%.1f TFLOP/s algorithmic)" % (ms, 2*27*32*32*48*136*240/ms/1e9))
ref64 = torch.nn.functional.conv3d(x[:, :, :8].double(), w.double(), padding=1)[:, :, 1:7]
print("full-shape slice err vs f64: %.3e" % (y[:, :, 1:7].double() - ref64).abs().max().item())

import sys, torch
sys.path.insert(0, "/root/repo")
import dcanet_amd
from dcanet_amd import ops
L = ops._L()
dev = "cuda"
x = torch.randn(1, 32, 48, 136, 240, device=dev); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
nb = L.dca_conv3d_x3_weight_bytes(32, 32)
wx = torch.empty(nb // 2, dtype=torch.int16, device=dev)
s = torch.cuda.current_stream().cuda_stream
L.dca_conv3d_x3_prep_weight(ops._ptr(w), ops._ptr(wx), 32, 32, 0, 0, s)
y = torch.empty(1, 32, 48, 136, 240, device=dev)
def run():
    rc = L.dca_conv3d_x3_forward(ops._ptr(x), ops._ptr(wx), ops._ptr(y), None, None, None, None, 1.0, 1, 32, 32, 48, 136, 240, s); assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
import os
print("DBG", os.environ.get("DCA_X3_DBG", "0"), "x3 32->32 @48x136x240: %.3f ms" % (e0.elapsed_time(e1) / 20))

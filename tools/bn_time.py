"""Launch times of the BatchNorm kernels at the layer shapes of the batch-4 training step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bn_time -- python tools/bn_time.py
(the summary names the kernels; one case per process invocation keeps them apart: BN_CASE=plain|pack|both|both_res)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
case = os.environ.get("BN_CASE", "plain")
N, C, D, H, W = [int(v) for v in os.environ.get("BN_SHAPE", "4,32,48,136,240").split(",")]
bn = torch.nn.BatchNorm3d(C).to(dev).train()
y = torch.randn(N, C, D, H, W, device=dev, requires_grad=True)
res = torch.randn(N, C, D, H, W, device=dev) if case == "both_res" else None
g = torch.randn(N, C, D, H, W, device=dev)
for _ in range(4):
    z = ops.bn_act(y, bn, 0.0, res_post=res, pack_out={"plain": False, "pack": True}.get(case, "both"))
    if ops._is_packed(z):
        # the packed result has no fp32 view: its gradient comes from a convolution; here: a packed gradient is not needed,
        # the backward of interest is that of the fp32 forms
        continue
    z.backward(g)
    y.grad = None
torch.cuda.synchronize()

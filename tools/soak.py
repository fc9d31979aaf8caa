"""Soak: N training steps at the BASELINE shape; loss must stay finite and decrease, memory must not grow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dcanet_amd.parallel import FlatGradBucket
dev = torch.device("cuda", 0)
m = bench.build_model(dev)
fL, fR, guid, gt = bench.make_inputs(1, 0, dev)
m.train(); fL.requires_grad_(); fR.requires_grad_()
params = bench.hot_params(m)
bucket = FlatGradBucket(params)
opt = torch.optim.Adam(params, lr=1e-4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mem0 = None
t0 = time.time()
for i in range(n):
    loss = bench.train_step(m, fL, fR, guid, gt, bucket, opt)
    if i % 50 == 0 or i == n - 1:
        l = float(loss)
        mem = torch.cuda.max_memory_allocated() / 2**30
        if i == 50: mem0 = mem
        print(f"step {i:4d} loss {l:.4f} max_mem {mem:.2f} GiB  {time.time() - t0:.1f}s", flush=True)
        assert l == l and abs(l) < 1e6, "loss not finite"
        if mem0 is not None: assert mem <= mem0 * 1.02 + 0.1, "memory grows"
print("soak ok")

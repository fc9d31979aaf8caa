"""torch.profiler CPU-side view of the training step (both the main and the autograd thread)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
m = bench.build_model(dev)
fL, fR, guid, gt = bench.make_inputs(1, 0, dev)
m.train(); fL.requires_grad_(); fR.requires_grad_()
from dcanet_amd.parallel import FlatGradBucket
params = bench.hot_params(m)
bucket = FlatGradBucket(params)
opt = torch.optim.Adam(params, lr=1e-3)
for _ in range(3): bench.train_step(m, fL, fR, guid, gt, bucket, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(3): bench.train_step(m, fL, fR, guid, gt, bucket, opt)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=32, max_name_column_width=48))

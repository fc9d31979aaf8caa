"""cProfile of the host side of the training step (where do the ~30 ms of Python per step go?)."""
import cProfile, pstats, sys, io, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
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
pr = cProfile.Profile()
pr.enable()
for _ in range(5): bench.train_step(m, fL, fR, guid, gt, bucket, opt)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])

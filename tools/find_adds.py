"""Which gradient accumulations does autograd still do with separate add launches in one training step?  (small shape)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
m = bench.build_model(dev).train()
fL, fR, guid, gt = bench.make_inputs(1, 0, dev)
fL.requires_grad_(); fR.requires_grad_()
from dcanet_amd.parallel import FlatGradBucket
bucket = FlatGradBucket(bench.hot_params(m))
for _ in range(2): bench.train_local(m, fL, fR, guid, gt, bucket)
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    bench.train_local(m, fL, fR, guid, gt, bucket)
rows = [e for e in prof.events() if e.name in ("aten::add", "aten::add_") and e.input_shapes and len(e.input_shapes[0]) >= 4]
import collections
c = collections.Counter((e.name, str(e.input_shapes[:2])) for e in rows)
for k, v in c.most_common(): print(v, k)

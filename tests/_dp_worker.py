"""Rank process of tests/test_parallel_cpu.py (gloo, CPU): prints the reduced flat gradient bucket."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd  # noqa: E402,F401
from dcanet_amd.parallel import FlatGradBucket, init_from_env, shard_batch  # noqa: E402

rank, _, world = init_from_env("gloo")
torch.manual_seed(0)
lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 1))
bucket = FlatGradBucket(lin.parameters())
x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10.0
bucket.zero()
lin(shard_batch(x, rank, world)).sum().backward()     # autograd assigns fresh .grad tensors
bucket.all_reduce_mean()                               # gather -> one all-reduce -> .grad = views of the bucket
assert all(bucket.flat.data_ptr() <= p.grad.data_ptr() < bucket.flat.data_ptr() + 4 * bucket.numel
           for p in lin.parameters())
bucket.zero()
lin(shard_batch(x, rank, world)).sum().backward()      # second step must not accumulate into the first
bucket.all_reduce_mean()
print("RESULT " + json.dumps(bucket.flat.tolist()), flush=True)
dist.barrier()
dist.destroy_process_group()

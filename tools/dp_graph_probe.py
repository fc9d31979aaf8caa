"""Where does a data-parallel GraphedTrainStep spend its time when two ranks share ONE GPU (gloo rehearsal)?
    DCA_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29533 tools/dp_graph_probe.py
Times, per rank and per step: graph A replay (forward + losses + backward + gather), the eager all-reduce, graph B replay
(Adam) -- each followed by a device synchronisation -- and the same step launched eagerly."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.H_IMG, bench.W_IMG, bench.MAXDISP = 64, 128, 32
from dcanet_amd.graph import GraphedTrainStep  # noqa: E402
from dcanet_amd.parallel import FlatGradBucket, init_from_env  # noqa: E402

rank, local, world = init_from_env()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
m = bench.build_model(dev).train()
fL, fR, guid, gt = bench.make_inputs(1, rank, dev)
fL.requires_grad_(); fR.requires_grad_()
params = bench.hot_params(m)
bucket = FlatGradBucket(params)
opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
sync = torch.cuda.synchronize


def timed(fn):
    sync(); t = time.perf_counter(); fn(); sync()
    return (time.perf_counter() - t) * 1e3


eager = []
for i in range(8):
    a = timed(lambda: bench.train_local(m, fL, fR, guid, gt, bucket))
    b = timed(bucket.reduce_flat)
    c = timed(opt.step)
    eager.append((a, b, c))
FORCE_B = "--force-b" in sys.argv          # single process, two graphs, a no-op in place of the all-reduce
SIDE = "--side-stream" in sys.argv         # ... or the collective's stream pattern: another stream waits for an event of
_side = torch.cuda.Stream()                # the step's stream, touches the gradient bucket [and copies it to the host and


def side_stream_op():                      # back, as gloo does], and the step's stream waits for that
    cur = torch.cuda.current_stream()
    _side.wait_stream(cur)
    with torch.cuda.stream(_side):
        if "--host" in sys.argv:
            h = bucket.flat.to("cpu", non_blocking=False)
            bucket.flat.copy_(h, non_blocking=False)
        else:
            bucket.flat.mul_(1.0)
    cur.wait_stream(_side)


g = GraphedTrainStep(lambda: bench.train_local(m, fL, fR, guid, gt, bucket), opt.step,
                     bucket.reduce_flat if world > 1 else (side_stream_op if SIDE else (lambda: None) if FORCE_B else None))
rows = []
for i in range(8):
    a = timed(g.graph_a.replay)
    b = timed(g.all_reduce) if g.graph_b is not None else 0.0
    c = timed(g.graph_b.replay) if g.graph_b is not None else 0.0
    rows.append((a, b, c))
whole = [timed(g) for _ in range(8)]


def run_unsynced(n, sync_after_a=False, sync_after_r=False):
    def body():
        for _ in range(n):
            g.graph_a.replay()
            if sync_after_a:
                sync()
            if g.graph_b is not None:
                g.all_reduce()
                if sync_after_r:
                    sync()
                g.graph_b.replay()
    return timed(body) / n


free = run_unsynced(10)
after_a = run_unsynced(10, sync_after_a=True)
after_r = run_unsynced(10, sync_after_r=True)
med = lambda xs: sorted(xs)[len(xs) // 2]
print(f"rank {rank}/{world}: eager  local {med([r[0] for r in eager]):7.2f}  all-reduce {med([r[1] for r in eager]):7.2f}  adam {med([r[2] for r in eager]):6.2f} ms\n"
      f"rank {rank}/{world}: graphs local {med([r[0] for r in rows]):7.2f}  all-reduce {med([r[1] for r in rows]):7.2f}  adam {med([r[2] for r in rows]):6.2f} ms"
      f"   whole graphed step, sync per step {med(whole):7.2f} ms\n"
      f"rank {rank}/{world}: 10 graphed steps back to back: no syncs {free:7.2f}  sync after A {after_a:7.2f}  sync after all-reduce {after_r:7.2f} ms/step",
      flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()

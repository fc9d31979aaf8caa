"""Rank process of tests/test_gpu_dp.py: the DCANet training step (hot path + convex up-sampler + both losses + backward +
flat-bucket all-reduce + Adam) of bench.py on this rank's shard of a seeded global batch, two steps.  World size 1 = the
single-process reference runs (WORLD_SIZE=1, DP_SHARD=i selects the shard).  All ranks share cuda:0 (gloo backend), so the
collective's numerics can be checked on a one-GPU box; on an 8-GPU node the same code runs over RCCL (backend "nccl")."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcanet_amd  # noqa: E402,F401
from dcanet_amd.models.gwcnet_dca_g import GwcNet  # noqa: E402
from dcanet_amd.models.loss import focal_loss, model_loss  # noqa: E402
from dcanet_amd.parallel import FlatGradBucket, init_from_env, shard_batch  # noqa: E402
from oracle import dcanet_oracle as O  # noqa: E402  (test infrastructure: the key-seeded weights)
from oracle.seeded import seeded_tensor  # noqa: E402

out_path = sys.argv[1]
D, H4, W4, GLOBAL_B = 32, 16, 32, 4
rank, local, world = init_from_env(os.environ.get("DCA_DIST_BACKEND", "gloo"))
shard = int(os.environ.get("DP_SHARD", rank))
nshards = int(os.environ.get("DP_NSHARDS", world))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

m = GwcNet(D, use_concat_volume=False)
m.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}), strict=True)
m = m.to(dev).train()
mods = [m.dres0, m.dres1, m.cva1, m.cva2, m.cva3, m.classif0, m.classif1, m.classif2, m.classif3, m.prop]
params = [p for mod in mods for p in mod.parameters()]
bucket = FlatGradBucket(params)
opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999))

fL = shard_batch(seeded_tensor("dp.fL", (GLOBAL_B, 320, H4, W4)), shard, nshards).to(dev).requires_grad_()
fR = shard_batch(seeded_tensor("dp.fR", (GLOBAL_B, 320, H4, W4)), shard, nshards).to(dev).requires_grad_()
guid = shard_batch(seeded_tensor("dp.guid", (GLOBAL_B, 64, H4, W4)), shard, nshards).to(dev)
gt = shard_batch(torch.rand(GLOBAL_B, 1, 4 * H4, 4 * W4, generator=torch.Generator().manual_seed(11)) * (D - 2.0) + 1.0,
                 shard, nshards).to(dev)

rec = {"world": world, "shard": shard}
for step in range(2):
    bucket.zero()
    fL.grad = fR.grad = None
    r = m.hot_path(fL, fR)
    pred4 = m.prop(guid, r["pred4_q"])
    mask = (gt < D) & (gt > 0)
    loss = focal_loss([r["pred0"], r["pred_dca1"], r["pred_dca2"], r["pred1"], r["pred2"]], gt, D, 5.0, False) \
        + model_loss([r["pred_dca3"], pred4], gt, mask)
    loss.backward()
    bucket.gather()
    rec[f"local{step}"] = bucket.flat.detach().cpu().clone()
    bucket.reduce_flat()
    rec[f"reduced{step}"] = bucket.flat.detach().cpu().clone()
    opt.step()
    rec[f"loss{step}"] = float(loss)
rec["params"] = torch.cat([p.detach().reshape(-1) for p in params]).cpu()
rec["exp_avg"] = torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in params]).cpu()
rec["exp_avg_sq"] = torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in params]).cpu()
rec["backend"] = dist.get_backend() if world > 1 else "none"
rec["dist_world"] = dist.get_world_size() if world > 1 else 1
torch.save(rec, out_path)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()

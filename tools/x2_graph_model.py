import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dcanet_amd
from dcanet_amd import ops
from dcanet_amd.graph import GraphedHotPath
from dcanet_amd.models.gwcnet_dca_g import GwcNet
from oracle import dcanet_oracle as O
from oracle.seeded import seeded_tensor
DEV = "cuda"


def load_seeded(module):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(O.seeded_state_dict(shapes), strict=True)
    return module


def words(tag, pool=None):
    pool = pool or getattr(ops._tls, "amax_pool", None)
    if pool is None:
        return
    n = pool[1]
    w = pool[0][:n * 64].view(torch.float32).view(n, 64).max(1).values.cpu().tolist()
    print(tag, "pool words used", n, "cap", pool[2], ["%.3g" % v for v in w], flush=True)


m = load_seeded(GwcNet(64, use_concat_volume=False)).to(DEV).eval()
fL, fR = seeded_tensor("gr.fL", (1, 320, 24, 40)).to(DEV), seeded_tensor("gr.fR", (1, 320, 24, 40)).to(DEV)
g = GraphedHotPath(m, fL, fR)
p3 = getattr(ops._tls, "amax_pool", None)
torch.cuda.synchronize()
words("after capture (not run yet)", p3)
if os.environ.get("REPLAY_FIRST"):
    got0 = g(fL, fR)["pred4_q"].clone()
    torch.cuda.synchronize()
    words("after replay 0", p3)
    print("replay0 finite", torch.isfinite(got0).all().item())
with torch.no_grad():
    want = m.hot_path(fL, fR)["pred4_q"].clone()
torch.cuda.synchronize()
words("eager after capture")
words("graph pool after the eager run", p3)
got = g(fL, fR)["pred4_q"]
torch.cuda.synchronize()
words("after replay", p3)
print("replay == eager", torch.equal(got, want), "finite", torch.isfinite(got).all().item())

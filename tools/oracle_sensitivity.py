"""How well-conditioned is the end-to-end train-mode gradient the GPU parity test gates?  Perturb the CPU oracle's
inputs by a relative eps (random signs) and report the rel-L2 change of d(loss)/d(fL) -- the yardstick for the gate
in tests/test_gpu_parity.py::test_golden_hot_path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dcanet_oracle as O
from oracle.seeded import seeded_tensor
torch.set_num_threads(8)
def run(fL, fR, sd):
    fL = fL.clone().requires_grad_(); fR = fR.clone().requires_grad_()
    r = O.hot_path(sd, fL, fR, 32, True)
    keys = ["pred0", "pred_dca1", "pred_dca2", "pred1", "pred2", "pred_dca3", "pred4_q"]
    loss = sum((r[k] * seeded_tensor(f"hot.g{i}", r[k].shape)).sum() for i, k in enumerate(keys))
    g, = torch.autograd.grad(loss, [fL])
    return g
sd = O.seeded_state_dict(O.hot_path_shapes(False))
fL, fR = seeded_tensor("hot.fL", (2, 320, 16, 32)), seeded_tensor("hot.fR", (2, 320, 16, 32))
g0 = run(fL, fR, O.clone_sd(sd))
for eps in (1e-7, 3e-7, 1e-6, 3e-6):
    out = []
    for seed in range(4):
        gen = torch.Generator().manual_seed(seed)
        p = lambda t: t * (1 + eps * (torch.randint(0, 2, t.shape, generator=gen).float() * 2 - 1))
        g1 = run(p(fL), p(fR), O.clone_sd(sd))
        out.append(((g1 - g0).norm() / g0.norm()).item())
    print("eps %.0e: rel-L2 change of gfL over 4 seeds: %s" % (eps, ", ".join("%.2e" % v for v in out)))

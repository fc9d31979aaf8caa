"""GPU parity of the SURVEY 8(f) kernels (csrc/heads2d.hip) against the CPU oracle and the reference's own outputs:
convex x4 up-sampling (PropgationNet_4x, submodule.py:366-373) and the stereo focal loss (loss.py:16-24, 168-247)."""
import numpy as np
import pytest
import torch

from oracle import dcanet_oracle as O
from oracle.seeded import seeded_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, tol=2e-5, name=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{name}: max err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("shape", [(2, 6, 10), (1, 1, 1), (3, 5, 67), (1, 34, 60)])
def test_convex_upsample_vs_oracle(shape):
    """incl. a single cell (all 8 neighbours padded) and widths that are not a multiple of the workgroup"""
    from dcanet_amd import ops
    B, h, w = shape
    mask = seeded_tensor(f"cvx.m{shape}", (B, 144, h, w)) * 2
    disp = seeded_tensor(f"cvx.d{shape}", (B, 1, h, w)) * 3 + 10
    gup = seeded_tensor(f"cvx.g{shape}", (B, 1, 4 * h, 4 * w))
    mc, dc = mask.clone().requires_grad_(), disp.clone().requires_grad_()
    ref = O.convex_upsample(mc, dc)
    rg = torch.autograd.grad((ref * gup).sum(), [mc, dc])
    mg, dg = mask.to(DEV).requires_grad_(), disp.to(DEV).requires_grad_()
    got = ops.convex_upsample4(mg, dg)
    close(got, ref, 1e-6, "up")
    gg = torch.autograd.grad((got * gup.to(DEV)).sum(), [mg, dg])
    close(gg[0], rg[0], 2e-6, "d mask"); close(gg[1], rg[1], 2e-6, "d disp")


def test_convex_upsample_is_a_convex_combination():
    """size-independent property at the BASELINE shape (136x240 cells): with a constant disparity map every interior
    output equals 4*disp exactly-ish (the softmax weights sum to 1), border cells are pulled towards 0 by the padding."""
    from dcanet_amd import ops
    mask = torch.randn(1, 144, 136, 240, device=DEV)
    up = ops.convex_upsample4(mask, torch.full((1, 1, 136, 240), 7.0, device=DEV))
    assert up.shape == (1, 1, 544, 960)
    assert (up[:, :, 4:-4, 4:-4] - 28.0).abs().max().item() < 1e-4
    assert up.max().item() <= 28.0 + 1e-4 and up.min().item() >= 0.0


def _loss_inputs():
    g = dict(np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "losses.npz")))
    gt = torch.from_numpy(g["gt"])
    ests = [torch.softmax(seeded_tensor(f"loss.e{i}", (2, 8, 8, 16)), 1) for i in range(5)]
    return g, gt, ests


@pytest.mark.parametrize("sparse", [False, True])
def test_focal_loss_matches_reference(sparse):
    """tests/golden/losses.npz = the reference's focal_loss(ests, gt, 32, 5.0, sparse) with ~15 % invalid ground truth"""
    from dcanet_amd.models.loss import focal_loss
    g, gt, ests = _loss_inputs()
    eg = [e.to(DEV).requires_grad_() for e in ests]
    fl = focal_loss(eg, gt.to(DEV), 32, 5.0, sparse)
    want = float(g["focal_sparse" if sparse else "focal"])
    assert abs(fl.item() - want) <= 1e-5 * max(1.0, abs(want)), (fl.item(), want)
    if not sparse:
        gr = torch.autograd.grad(fl, eg)
        close(gr[0], g["gfocal0"], 1e-5, "d est0"); close(gr[4], g["gfocal4"], 1e-5, "d est4")


@pytest.mark.parametrize("case", [(2, 48, 20, 36, 4), (1, 24, 9, 13, 8), (1, 192, 8, 12, 1), (3, 16, 5, 7, 2)])
def test_focal_loss_vs_oracle(case):
    """other bin counts (incl. K = 192 > 64: the one-wave variant), odd sizes, un-pooled ground truth (scale 1), a batch
    with no valid pixel at all, and StereoFocalLoss called the way train_kitti.py:110 calls it"""
    from dcanet_amd.models.loss import StereoFocalLoss, focal_loss
    B, K, H, W, s = case
    D = K * s
    gt = torch.rand(B, 1, H * s, W * s, generator=torch.Generator().manual_seed(K)) * (D + 8.0) - 4.0
    ests = [seeded_tensor(f"fl.{case}.{i}", (B, K, H, W)) * (1 + i) for i in range(3)]
    ec = [e.clone().requires_grad_() for e in ests]
    ref = O.focal_loss(ec, gt, D, 5.0, False)
    rg = torch.autograd.grad(ref, ec)
    eg = [e.to(DEV).requires_grad_() for e in ests]
    got = focal_loss(eg, gt.to(DEV), D, 5.0, False)
    close(got, ref, 1e-5, "focal")
    gg = torch.autograd.grad(got, eg)
    for i in range(3):
        close(gg[i], rg[i], 1e-5, f"d est{i}")
    ev = StereoFocalLoss(max_disp=D, focal_coefficient=2.0, sparse=True)
    one = 5 * ev(eg[0], gt.to(DEV), variance=1)
    close(one, 5 * O.stereo_focal_loss_level(ests[0], gt, D, 2.0, True), 1e-5, "StereoFocalLoss.__call__")
    none_valid = torch.full_like(gt, -1.0)
    z = focal_loss(eg, none_valid.to(DEV), D, 5.0, False)
    assert z.item() == 0.0 and O.focal_loss(ests, none_valid, D, 5.0, False).item() == 0.0
    assert all(t.abs().max().item() == 0.0 for t in torch.autograd.grad(z, eg))


def test_training_losses_full_size_properties():
    """BASELINE shape (B,48,136,240) x 5 levels: finite, positive, gradient of each level sums to ~0 over the disparity
    axis (softmax Jacobian) and scales with the level weight."""
    from dcanet_amd.models.loss import focal_loss
    gt = torch.rand(1, 1, 544, 960, device=DEV) * 190 + 1
    ests = [torch.softmax(torch.randn(1, 48, 136, 240, device=DEV), 1).requires_grad_() for _ in range(5)]
    fl = focal_loss(ests, gt, 192, 5.0, False)
    assert torch.isfinite(fl) and fl.item() > 0
    gr = torch.autograd.grad(fl, ests)
    for g_ in gr:
        assert g_.sum(1).abs().max().item() < 1e-9 * 48 + 1e-10
    r = gr[4].norm() / gr[0].norm()
    assert 0.5 < r.item() < 20.0

"""Mirror of the reference's models/loss.py for the two losses main_dca.py:132-133 uses.

Host-side PyTorch code (works on any device): it sits AFTER the hot path and is small
next to it.  Vectorised restatement of StereoFocalLoss.loss_per_level / LaplaceDisp2Prob (loss.py:117-128,
206-240) and model_loss (loss.py:6-14); quirks of the reference are reproduced, not fixed (the estimates
handed in are already softmax outputs and get log_softmax-ed again; the mean runs over all pixels)."""
import torch
import torch.nn.functional as F


def model_loss(disp_ests, disp_gt, mask):
    """reference loss.py:6-14: 1.8*SmoothL1(est0[mask]) + 2.1*SmoothL1(est1[mask]), mean over the masked pixels.

    Written as sum(mask * loss) / count(mask) instead of boolean indexing: same value and gradient (0/0 = NaN for an
    empty mask, like the mean of an empty selection), but no `nonzero` -- boolean indexing synchronises the host with
    the GPU in the middle of the training step, and everything after it is launched into an empty queue."""
    weights = [1.8, 2.1]
    assert len(weights) == len(disp_ests)
    m = mask.to(disp_gt.dtype)
    count = m.sum()
    return sum(w * (F.smooth_l1_loss(est, disp_gt, reduction="none") * m).sum() / count
               for est, w in zip(disp_ests, weights))


def _focal_target(gt, H, W, maxdisp, focal_coefficient, sparse, dtype, device):
    """everything of a level that does not depend on the estimate: the Laplace target distribution of the (pooled)
    ground truth, times the focal weight, times the validity mask (reference loss.py:117-128, 206-240)"""
    gt = gt.view(gt.shape[0], 1, gt.shape[-2], gt.shape[-1]) if gt.dim() != 4 else gt
    scale = 1.0
    sgt = gt
    if gt.shape[-2] != H or gt.shape[-1] != W:
        scale = gt.shape[-1] / (W * 1.0)
        pool = F.adaptive_max_pool2d if sparse else F.adaptive_avg_pool2d
        sgt = pool(gt / scale, (H, W))
    nd = int(maxdisp / scale)
    mask = ((sgt > 0) & (sgt < nd)).to(dtype)
    mgt = sgt * mask
    inner = ((mgt > 0) & (mgt < nd - 1)).to(dtype)          # Disp2Prob.getProb, loss.py:87-90
    index = torch.arange(0, nd, dtype=dtype, device=device).view(1, nd, 1, 1)
    prob = F.softmax(-torch.abs(index - mgt * inner), dim=1) * inner + 1e-40
    prob = prob * (mask.sum() >= 1.0).to(dtype)            # "no valid point" -> zero target (loss.py:224-227)
    weight = (1.0 - prob).pow(-focal_coefficient)
    return prob, weight, mask


def _focal_level(est, target):
    prob, weight, mask = target
    logp = F.log_softmax(est, dim=1)
    return -((prob * logp) * weight * mask).sum(dim=1, keepdim=True).mean()


def focal_loss(disp_ests, disp_gt, maxdisp, focal_coefficient, sparse):
    """reference loss.py:16-24 (weights [0.5,0.7,1.0,1.2,1.5], silently truncating like zip does).  The target side of
    a level depends only on the ground truth and the level's resolution, so levels of equal resolution (all five in
    DCANet) share one target instead of re-pooling / re-softmaxing the ground truth five times."""
    weights = [0.5, 0.7, 1.0, 1.2, 1.5]
    targets = {}
    total = 0
    for est, w in zip(disp_ests, weights):
        key = (est.shape[-2], est.shape[-1], est.dtype, est.device)
        if key not in targets:
            targets[key] = _focal_target(disp_gt, est.shape[-2], est.shape[-1], maxdisp, focal_coefficient, sparse,
                                         est.dtype, est.device)
        total = total + w * _focal_level(est, targets[key])
    return total

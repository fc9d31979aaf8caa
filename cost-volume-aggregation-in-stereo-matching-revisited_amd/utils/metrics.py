"""Disparity metrics with the reference's semantics (utils/metrics.py:22-70; main_dca.py:200-207 uses the same
quantities inline): computed per image over the masked pixels, images whose mask covers < 10 % of their positive
ground truth are skipped, the batch value is the mean over the remaining images (0 if none).  Inputs (B,H,W)."""
import torch


def _per_image(fn, D_ests, D_gts, masks, *args):
    assert D_ests.dim() == 3 and D_ests.shape == D_gts.shape == masks.shape
    vals = []
    with torch.no_grad():
        for est, gt, m in zip(D_ests, D_gts, masks):
            if m.float().mean() / (gt > 0).float().mean() < 0.1:
                continue
            vals.append(fn(est[m], gt[m], *args))
    if not vals:
        return torch.tensor(0, dtype=torch.float32, device=D_gts.device)
    return torch.stack(vals).mean()


def EPE_metric(D_ests, D_gts, masks):
    """end-point error: mean |est - gt| over the mask"""
    return _per_image(lambda e, g: (e - g).abs().mean(), D_ests, D_gts, masks)


def D1_metric(D_ests, D_gts, masks):
    """KITTI D1: share of masked pixels with error > 3 px and > 5 % of the ground truth"""
    return _per_image(lambda e, g: (((e - g).abs() > 3) & ((e - g).abs() / g.abs() > 0.05)).float().mean(),
                      D_ests, D_gts, masks)


def Thres_metric(D_ests, D_gts, masks, thres):
    """share of masked pixels with error > thres"""
    assert isinstance(thres, (int, float))
    return _per_image(lambda e, g, t: ((e - g).abs() > t).float().mean(), D_ests, D_gts, masks, thres)

"""Mirror of the reference's models/loss.py for the symbols its scripts import: `focal_loss`, `model_loss`
(main_dca.py:14,132-133) and `StereoFocalLoss` (train_kitti.py:16,101-110).

The stereo focal loss runs as ONE HIP kernel per direction for all levels of a resolution (csrc/heads2d.hip: log-softmax,
Laplace target, focal weight, masks and the mean fused; the reference materialises five (B,48,h,w) temporaries per
level).  Quirks of the reference are reproduced, not fixed: the estimates handed in by GwcNet are already softmax
outputs and get log_softmax-ed again; the mean runs over all pixels, valid or not.  Like every kernel of this package
the focal loss needs ROCm tensors: there is no CPU fallback."""
import torch
import torch.nn.functional as F

from ._bootstrap import ensure as _ensure

ops = _ensure().ops


def model_loss(disp_ests, disp_gt, mask):
    """reference loss.py:6-14: 1.8*SmoothL1(est0[mask]) + 2.1*SmoothL1(est1[mask]), mean over the masked pixels.

    Written with torch.where instead of boolean indexing: same value and gradient for finite inputs, no `nonzero` (a
    host<->GPU sync in the middle of the training step), and -- unlike a multiplication by the mask -- a non-finite
    estimate or ground truth at a masked-OUT pixel stays out of the sum, as with the reference's `est[mask]`.  An empty
    mask gives 0/0 = NaN like the mean of an empty selection.

    Data parallel note: every rank takes its own masked mean and the gradients are averaged over ranks; that equals
    nn.DataParallel's global masked mean only when all ranks hold the same number of valid pixels (true for dense
    ground truth).  `model_loss_parts` returns (numerator, count) for callers that all-reduce both."""
    num, count = model_loss_parts(disp_ests, disp_gt, mask)
    return num / count


def model_loss_parts(disp_ests, disp_gt, mask):
    weights = [1.8, 2.1]
    assert len(weights) == len(disp_ests)
    zero = torch.zeros((), dtype=disp_gt.dtype, device=disp_gt.device)
    num = sum(w * torch.where(mask, F.smooth_l1_loss(est, disp_gt, reduction="none"), zero).sum()
              for est, w in zip(disp_ests, weights))
    return num, mask.sum().to(disp_gt.dtype)


def _pooled_gt(gt, H, W, sparse):
    """loss_per_level's ground-truth scaling (loss.py:208-215): gt / scale pooled to the level's resolution;
    scale = W_gt / W.  Returns (pooled gt (B,1,H,W), scale)."""
    if gt.dim() == 2:
        gt = gt.view(1, 1, *gt.shape)
    elif gt.dim() == 3:
        gt = gt.view(gt.shape[0], 1, gt.shape[1], gt.shape[2])
    if gt.shape[-2] == H and gt.shape[-1] == W:
        return gt, 1.0
    scale = gt.shape[-1] / (W * 1.0)
    pool = F.adaptive_max_pool2d if sparse else F.adaptive_avg_pool2d
    return pool(gt / scale, (H, W)), scale


class StereoFocalLoss(object):
    """reference loss.py:168-247, same constructor and call signature.  `variance` and `dilation` are accepted and, as in
    the reference's LaplaceDisp2Prob.calProb (loss.py:122-126), not used."""

    def __init__(self, max_disp=192, start_disp=0, dilation=1, weights=None, focal_coefficient=0.0, sparse=False):
        if start_disp != 0:
            raise NotImplementedError("StereoFocalLoss: only start_disp = 0 (every call site of the reference)")
        self.max_disp = max_disp
        self.start_disp = start_disp
        self.dilation = dilation
        self.weights = weights
        self.focal_coefficient = focal_coefficient
        self.sparse = sparse
        self.scale_func = F.adaptive_max_pool2d if sparse else F.adaptive_avg_pool2d

    def loss_levels(self, ests, weights, gtDisp):
        """sum_l weights[l] * loss_per_level(ests[l]) for estimates of one resolution: one kernel launch."""
        N, C, H, W = ests[0].shape
        gt, scale = _pooled_gt(gtDisp, H, W, self.sparse)
        if int(self.max_disp / scale) != C:
            raise RuntimeError(f"StereoFocalLoss: the estimate has {C} disparity bins but max_disp / scale = "
                               f"{int(self.max_disp / scale)} (the reference's broadcast would fail as well)")
        return ops.focal_loss_levels(list(ests), gt, list(weights), self.focal_coefficient)

    def loss_per_level(self, estCost, gtDisp, variance=1.0, dilation=1):
        return self.loss_levels([estCost], [1.0], gtDisp)

    def __call__(self, estCost, gtDisp, variance):
        return self.loss_per_level(estCost, gtDisp, variance)


def focal_loss(disp_ests, disp_gt, maxdisp, focal_coefficient, sparse):
    """reference loss.py:16-24 (weights [0.5,0.7,1.0,1.2,1.5], silently truncating like zip does).  Levels of equal shape
    (all five in DCANet) go through one kernel launch that computes the ground-truth side once."""
    weights = [0.5, 0.7, 1.0, 1.2, 1.5]
    ev = StereoFocalLoss(max_disp=maxdisp, focal_coefficient=focal_coefficient, sparse=sparse)
    groups = {}
    for est, w in zip(disp_ests, weights):
        groups.setdefault(tuple(est.shape), ([], []))
        groups[tuple(est.shape)][0].append(est)
        groups[tuple(est.shape)][1].append(w)
    total = 0
    for ests, ws in groups.values():
        for i in range(0, len(ests), 8):
            total = total + ev.loss_levels(ests[i:i + 8], ws[i:i + 8], disp_gt)
    return total

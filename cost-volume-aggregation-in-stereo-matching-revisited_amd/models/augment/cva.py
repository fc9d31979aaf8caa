"""Mirror of the reference's models/augment/cva.py (Multi_Aggregation, cva) on HIP kernels."""
import torch
import torch.nn as nn

from .._bootstrap import ensure as _ensure
from ..submodule import ConvBn3d, ConvBnReLU3d, conv3d_plain, convbn_3d
from .semantic_level import SemanticLevelContext

ops = _ensure().ops


class _DeconvBn3d(nn.Sequential):
    """Sequential(ConvTranspose3d(k3, s2, p1, op1, bias=False), BatchNorm3d) -- keys '0.weight', '1.*'."""

    def __init__(self, cin, cout):
        super().__init__(nn.ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm3d(cout))

    def forward(self, x, slope=1.0, res_pre=None, res_post=None, pack_out=False):
        return ops.convbn3d(x, self[0], self[1], slope, res_pre, res_post, pack_out=pack_out)


class Multi_Aggregation(nn.Module):
    """reference cva.py:13-31: conv s2 -> conv -> deconv s2, 1x1x1 skip, ReLU(sum)."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = ConvBnReLU3d(in_channels, in_channels * 2, 3, 2, 1)
        self.conv2 = ConvBnReLU3d(in_channels * 2, in_channels * 2, 3, 1, 1)
        self.conv3 = _DeconvBn3d(in_channels * 2, in_channels)
        self.redir = convbn_3d(in_channels, in_channels, kernel_size=1, stride=1, pad=0)

    def forward(self, x, res_post=None):
        # conv1 (stride 2) and redir (1x1x1) read the same x: one autograd node, so their two gradients of x are summed in
        # the second backward-data launch instead of by a separate accumulation pass (ops.convbn3d_pair)
        # (c1 has one consumer, the 3x3x3 convolution conv2: packed px2 operand in training)
        c1, skip = ops.convbn3d_pair(x, self.conv1[0][0], self.conv1[0][1], 0.0, self.redir[0], self.redir[1], 1.0,
                                     pack_a=True)
        c2 = self.conv2(c1)
        # relu(conv3(c2) + redir(x)) [+ res_post, fused: the caller's `cost0 + augmented_cost`]
        # (the block's output is read by the next classifier head's 3x3x3 convolution -- through a packed twin -- and by the
        # next block's pooling / `fuse` as fp32)
        return self.conv3(c2, slope=0.0, res_pre=skip, res_post=res_post, pack_out="both")


class _Downsample(nn.Sequential):
    """Sequential(AvgPool3d(3,2,1), convbn_3d(32,32,3,1,1), ReLU) -- keys '1.0.weight', '1.1.*'."""

    def __init__(self):
        super().__init__(nn.AvgPool3d((3, 3, 3), stride=2, padding=1), ConvBn3d(32, 32, 3, 1, 1),
                         nn.ReLU(inplace=True))

    def forward(self, x):
        return self[1](ops.avg_pool3d_k3s2p1(x), slope=0.0)


class _Classify(nn.Sequential):
    """Sequential(convbn_3d, ReLU, Conv3d(C,1,3,p1)) -- keys '0.0.weight', '0.1.*', '2.weight'."""

    def __init__(self, c):
        super().__init__(ConvBn3d(c, c, 3, 1, 1), nn.ReLU(inplace=True),
                         nn.Conv3d(c, 1, kernel_size=3, padding=1, stride=1, bias=False))

    def forward(self, x, alias=False):
        """alias=True: returns (logits, x') with x' = x for the other consumers of x (ops._Conv3d.forward, `alias`)"""
        if alias:
            h, xa = self[0](x, slope=0.0, alias=True)
            return conv3d_plain(h, self[2]), xa
        h = self[0](x, slope=0.0)
        if h.dtype != torch.float32:            # reduced-precision inference: 2-byte features, fp32 logits
            return ops.conv3d_c1_lp(h, self[2].weight)
        return conv3d_plain(h, self[2])


class cva(nn.Module):
    """reference cva.py:33-72.  forward(cost_volume) -> (prob_volume.unsqueeze(1), augmented_cost)."""

    def __init__(self, max_disp, in_channel, downsample=True):
        super().__init__()
        self.max_disp = max_disp
        self.channel = in_channel
        if downsample:
            self.downsample = _Downsample()
        self.slc_net = SemanticLevelContext(feats_channels=self.channel, transform_channels=self.channel,
                                            concat_input=True)
        self.classify = _Classify(self.channel)
        self.fuse = nn.Sequential(ConvBn3d(64, 32, 1, 1, 0))
        self.cost_agg = Multi_Aggregation(self.channel)

    def _forward_lp(self, x, res_post):
        """Reduced-precision inference (ops.reduced_precision): the 1/4-resolution tensors (x, aug, fuse / redir / deconv
        outputs) are stored in the 2-byte type, the 1/8-resolution interior (pooled volume, logits, context injection,
        attention) stays fp32 storage with single-product convolutions."""
        lp = x.dtype
        aff = lambda bn: (lambda st, C: (st[2 * C:3 * C], st[3 * C:]))(ops.bn_eval_affine(bn), bn.num_features)
        cost_down = self.downsample[1](ops.avg_pool3d_lp(x), slope=0.0)
        prob_volume = self.classify(cost_down).squeeze(1)
        aug = ops.trilinear_up2_lp(self.slc_net(cost_down, prob_volume), lp)
        sc, sh = aff(self.fuse[0][1])
        fused = ops.conv1x1_lp(aug, self.fuse[0][0].weight, lp, x2=x, scale=sc, shift=sh, slope=1.0)
        agg = self.cost_agg
        sc, sh = aff(agg.conv1[0][1])
        c1 = ops.conv3d_s2_lp(fused, agg.conv1[0][0].weight, sc, sh, 0.0)
        c2 = agg.conv2(c1)
        sc, sh = aff(agg.redir[1])
        skip = ops.conv1x1_lp(fused, agg.redir[0].weight, lp, scale=sc, shift=sh, slope=1.0)
        sc, sh = aff(agg.conv3[1])
        out = ops.deconv3d_lp(c2, agg.conv3[0].weight, lp, sc, sh, 0.0, res_pre=skip, res_post=res_post)
        return prob_volume.unsqueeze(1), out

    def forward(self, cost_volume, downsample=True, res_post=None):
        if cost_volume.dtype != torch.float32:
            if not downsample:
                raise NotImplementedError("reduced-precision path: only the down-sampling form used by GwcNet")
            return self._forward_lp(cost_volume, res_post)
        if downsample:
            # pooled input + the input itself for `fuse` from one autograd node (ops._PoolFork): the two gradients of the
            # cost volume are summed inside the pooling backward kernel
            if res_post is cost_volume:      # `cost0 + augmented_cost`: a third reader of the same tensor
                pooled, cost_volume, res_post = ops.avg_pool3d_fork(cost_volume, third=True)
            else:
                pooled, cost_volume = ops.avg_pool3d_fork(cost_volume)
            cost_down = self.downsample[1](pooled, slope=0.0)
            prob_volume = self.classify(cost_down).squeeze(1)
            aug_down = self.slc_net(cost_down, prob_volume)
            aug = ops.trilinear_upsample(aug_down, 2)
        else:
            prob_volume = self.classify(cost_volume).squeeze(1)
            aug = self.slc_net(cost_volume, prob_volume)
        # fuse(cat([aug, cost_volume], 1)) without materialising the 64-channel concat
        aug = self.fuse[0](aug, x2=cost_volume)
        aug = self.cost_agg(aug, res_post=res_post)
        return prob_volume.unsqueeze(1), aug

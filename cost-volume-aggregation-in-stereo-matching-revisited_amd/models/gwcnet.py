"""Mirror of the reference's models/gwcnet.py: the baseline GwcNet (three stacked 3D hourglasses) on HIP kernels."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._bootstrap import ensure as _ensure
from .augment.cva import _Classify, _DeconvBn3d
from .gwcnet_dca_g import _Dres0, _Dres1, feature_extraction as _feature_extraction
from .submodule import ConvBnReLU3d, build_concat_volume, build_gwc_volume, convbn_3d

ops = _ensure().ops


class feature_extraction(_feature_extraction):
    """reference gwcnet.py:11-65 (same network as gwcnet_dca_g's; only the default of concat_feature differs)."""

    def __init__(self, concat_feature=True, concat_feature_channel=12):
        super().__init__(concat_feature=concat_feature, concat_feature_channel=concat_feature_channel)


class hourglass(nn.Module):
    """reference gwcnet.py:67-104 (identical copy at gwcnet_dca_g.py:69-106): 2-level 3D encoder/decoder."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        self.conv1 = ConvBnReLU3d(c, c * 2, 3, 2, 1)
        self.conv2 = ConvBnReLU3d(c * 2, c * 2, 3, 1, 1)
        self.conv3 = ConvBnReLU3d(c * 2, c * 4, 3, 2, 1)
        self.conv4 = ConvBnReLU3d(c * 4, c * 4, 3, 1, 1)
        self.conv5 = _DeconvBn3d(c * 4, c * 2)
        self.conv6 = _DeconvBn3d(c * 2, c)
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)

    def forward(self, x):
        conv2 = self.conv2(self.conv1(x))
        conv4 = self.conv4(self.conv3(conv2))
        conv5 = self.conv5(conv4, slope=0.0, res_pre=self.redir2(conv2))   # relu(conv5 + redir2(conv2))
        return self.conv6(conv5, slope=0.0, res_pre=self.redir1(x))        # relu(conv6 + redir1(x))


class GwcNet(nn.Module):
    """reference gwcnet.py:107-249.  forward(left, right, disp_true_down=None): training -> [pred0..pred3], each
    (B,1,H,W) full-resolution disparity; eval -> the reference's t-SNE visualisation volume (gwcnet.py:185-189,
    241-249), reproduced as is."""

    def __init__(self, maxdisp, use_concat_volume=False):
        super().__init__()
        self.maxdisp = maxdisp
        self.use_concat_volume = use_concat_volume
        self.num_groups = 40
        if self.use_concat_volume:
            self.concat_channels = 12
            self.feature_extraction = feature_extraction(concat_feature=True,
                                                         concat_feature_channel=self.concat_channels)
        else:
            self.concat_channels = 0
            self.feature_extraction = feature_extraction(concat_feature=False)
        self.dres0 = _Dres0(self.num_groups + self.concat_channels * 2)
        self.dres1 = _Dres1()
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        self.classif0 = _Classify(32)
        self.classif1 = _Classify(32)
        self.classif2 = _Classify(32)
        self.classif3 = _Classify(32)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.Conv3d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def vis_tsne1(self, out):
        """reference gwcnet.py:185-189 (hard-coded pooling sizes of the paper's figure)."""
        cost = self.classif2(out)
        cost = F.adaptive_avg_pool3d(cost[:, :, :, 2:, :], (48 // 2, 134 // 2, 240 // 2))
        return cost.squeeze(1)

    def hot_path(self, gwc_left, gwc_right, concat_left=None, concat_right=None):
        with ops.batched_bn_counters():   # one multi-tensor add for all BatchNorm step counters
            return self._hot_path(gwc_left, gwc_right, concat_left, concat_right)

    def _hot_path(self, gwc_left, gwc_right, concat_left=None, concat_right=None):
        """reference gwcnet.py:194-238 from the 1/4-res features."""
        d = self.maxdisp // 4
        volume = build_gwc_volume(gwc_left, gwc_right, d, self.num_groups)
        if self.use_concat_volume:
            volume = torch.cat((volume, build_concat_volume(concat_left, concat_right, d)), 1)
        cost0 = self.dres1(self.dres0(volume))                     # dres1(cost0) + cost0 fused
        out1 = self.dres2(cost0)
        out2 = self.dres3(out1)
        out3 = self.dres4(out2)
        if not self.training:
            return {"out2": out2}
        # classif -> trilinear x4 to [maxdisp, H, W] -> softmax -> regression, fused per head
        heads = [(self.classif0, cost0), (self.classif1, out1), (self.classif2, out2), (self.classif3, out3)]
        return {"preds": [ops.up_softargmin(cls(t).squeeze(1), 4) for cls, t in heads], "out2": out2}

    def forward(self, left, right, disp_true_down=None):
        assert left.shape[2] % 4 == 0 and left.shape[3] % 4 == 0 and self.maxdisp % 4 == 0
        fl = self.feature_extraction(left)
        fr = self.feature_extraction(right)
        r = self.hot_path(fl["gwc_feature"], fr["gwc_feature"], fl.get("concat_feature"), fr.get("concat_feature"))
        if self.training:
            return r["preds"]
        return self.vis_tsne1(r["out2"])


def GwcNet_G(d):
    return GwcNet(d, use_concat_volume=False)


def GwcNet_GC(d):
    return GwcNet(d, use_concat_volume=True)

"""Mirror of the reference's models/gwcnet_dca_g.py: DCANet (GwcNet + 3 DCA blocks)."""
import math

import torch
import torch.nn as nn

from ._bootstrap import ensure as _ensure
from .augment.cva import cva, _Classify
from .submodule import (BasicBlock, ConvBnReLU3d, ConvBn3d, Guidance, PropgationNet_4x, build_concat_volume,
                        build_gwc_volume, convbn, disparity_regression)  # noqa: F401

ops = _ensure().ops


class _Features(dict):
    """The extractor's result: `gwc_segments` = (l2, l3, l4), which the fused cost-volume kernel reads in place, and the
    reference's `gwc_feature` = torch.cat((l2, l3, l4), 1) (gwcnet_dca_g.py:60) built only if somebody asks for it."""

    def __missing__(self, key):
        if key != "gwc_feature":
            raise KeyError(key)
        self[key] = torch.cat(self["gwc_segments"], dim=1)
        return self[key]


class feature_extraction(nn.Module):
    """reference gwcnet_dca_g.py:13-66 -- 2D backbone, caller of the hot path (stays on PyTorch-ROCm)."""

    def __init__(self, concat_feature=False, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        self.inplanes = 32
        self.firstconv = nn.Sequential(convbn(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._make_layer(BasicBlock, 32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(BasicBlock, 64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(BasicBlock, 128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(BasicBlock, 128, 3, 1, 1, 2)
        if self.concat_feature:
            self.lastconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, concat_feature_channel, kernel_size=1, padding=0, stride=1,
                                                    bias=False))

    def _make_layer(self, block, planes, blocks, stride, pad, dilation):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1,
                                                 stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, pad, dilation)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, 1, None, pad, dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.layer1(self.firstconv(x))
        l2 = self.layer2(x)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        out = _Features(gwc_segments=(l2, l3, l4))
        if self.concat_feature:
            out["concat_feature"] = self.lastconv(out["gwc_feature"])
        return out


class _Dres0(nn.Sequential):
    """Sequential(convbn_3d, ReLU, convbn_3d, ReLU) -- reference gwcnet_dca_g.py:141-144."""

    def __init__(self, cin):
        super().__init__(ConvBn3d(cin, 32, 3, 1, 1), nn.ReLU(inplace=True), ConvBn3d(32, 32, 3, 1, 1),
                         nn.ReLU(inplace=True))

    def forward(self, x):
        # the first layer's only consumer is the second convolution: packed px2 operand in training (ops.convbn3d)
        # the result (cost0 before the residual block) is read by dres1's first convolution AND as its residual: fp32 plus
        # a packed twin for the convolution
        return self[2](self[0](x, slope=0.0, pack_out=True), slope=0.0, pack_out="both")


class _Dres1(nn.Sequential):
    """Sequential(convbn_3d, ReLU, convbn_3d) -- reference gwcnet_dca_g.py:146-148; the `+ cost0` of :225 is
    fused into the second conv's epilogue."""

    def __init__(self):
        super().__init__(ConvBn3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), ConvBn3d(32, 32, 3, 1, 1))

    def forward(self, x):
        # the residual reads x through the first convolution's alias output: its gradient is added inside that convolution's
        # backward-data launch (ops._Conv3d.forward, `alias`)
        h, xa = self[0](x, slope=0.0, alias=True, pack_out=True)
        # cost0: read by classif0's / classif3's convolution (packed twin) and by the pooling, `fuse` and residual adds (fp32)
        return self[2](h, slope=1.0, res_post=xa, pack_out="both")


class GwcNet(nn.Module):
    """reference gwcnet_dca_g.py:126-282.  Same constructor, attributes, state-dict keys and outputs;
    `disp_true` is optional because the reference's own scripts call `model(imgL, imgR)`
    (main_dca.py:131,169) while its forward declares a third, unused, argument."""

    def __init__(self, maxdisp, use_concat_volume=True):
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
        self.cva1 = cva(self.maxdisp, 32, downsample=True)
        self.cva2 = cva(self.maxdisp, 32, downsample=True)
        self.cva3 = cva(self.maxdisp, 32, downsample=True)
        self.classif0 = _Classify(32)
        self.classif1 = _Classify(32)
        self.classif2 = _Classify(32)
        self.classif3 = _Classify(32)
        self.guidance = Guidance(64)
        self.prop = PropgationNet_4x(64)
        # reference init, gwcnet_dca_g.py:173-185 (ConvTranspose3d keeps the PyTorch default)
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

    # ---- the hot path proper: 1/4-res features -> 1/4-res disparity (+ training heads)
    def hot_path(self, gwc_left, gwc_right, concat_left=None, concat_right=None):
        with ops.batched_bn_counters():   # one multi-tensor add for all BatchNorm step counters
            return self._hot_path(gwc_left, gwc_right, concat_left, concat_right)

    def _hot_path(self, gwc_left, gwc_right, concat_left=None, concat_right=None):
        """reference gwcnet_dca_g.py:216-239 (+ :244-275 when training).  Returns a dict with `pred4_q`
        (B,1,H/4,W/4) in 1/4-res pixels, `prob_volume2` and, in training mode, the auxiliary heads."""
        d = self.maxdisp // 4
        lp = ops._lp_dtype()
        if lp is not None and (self.training or torch.is_grad_enabled()):
            raise RuntimeError("ops.reduced_precision is inference only: call the model in eval mode under torch.no_grad()")
        # gwc (+ concat) volume in ONE tensor, from the extractor's l2/l3/l4 maps in place when they are handed over as
        # a tuple (no torch.cat of the features, no torch.cat of the two volumes: gwcnet_dca_g.py:60,217-220)
        volume = ops.cost_volume(gwc_left, gwc_right, d, self.num_groups,
                                 concat_left if self.use_concat_volume else None,
                                 concat_right if self.use_concat_volume else None,
                                 out_dtype=torch.float32 if lp is None else lp)
        cost0 = self.dres0(volume)
        cost0 = self.dres1(cost0)                               # dres1(cost0) + cost0
        heads = self.training and lp is None
        if heads:
            # the training heads classif0-2 read cost0 / out1 / out2 FIRST and hand them on as their first convolution's
            # alias output: the later consumers' gradient is added inside that convolution's backward-data launch instead of
            # by autograd's accumulation pass (ops._Conv3d.forward, `alias`); same values, other launch order
            logits0, cost0 = self.classif0(cost0, alias=True)
        prob_volume1, out1 = self.cva1(cost0, res_post=cost0)   # cost0 + augmented_cost
        if heads:
            logits1, out1 = self.classif1(out1, alias=True)
        prob_volume2, out2 = self.cva2(out1)
        if heads:
            logits2, out2 = self.classif2(out2, alias=True)
        prob_volume3, out3 = self.cva3(out2)
        logits3 = self.classif3(out3).squeeze(1)
        res = {"pred4_q": ops.softargmin(logits3), "prob_volume2": prob_volume2}
        if self.training:
            res["pred0"] = ops.softmax_dim1(logits0.squeeze(1))
            res["pred_dca1"] = ops.softmax_dim1(ops.trilinear_upsample(prob_volume1, 2).squeeze(1))
            res["pred_dca2"] = ops.softmax_dim1(ops.trilinear_upsample(prob_volume2, 2).squeeze(1))
            res["pred_dca3"] = ops.up_softargmin(prob_volume3.squeeze(1), 8)
            res["pred1"] = ops.softmax_dim1(logits1.squeeze(1))
            res["pred2"] = ops.softmax_dim1(logits2.squeeze(1))
        return res

    def forward(self, left, right, disp_true=None):
        features_left = self.feature_extraction(left)
        features_right = self.feature_extraction(right)
        guidance = self.guidance(left)
        r = self.hot_path(features_left["gwc_segments"], features_right["gwc_segments"],
                          features_left.get("concat_feature"), features_right.get("concat_feature"))
        pred4 = self.prop(guidance["g"], r["pred4_q"])
        if self.training:
            return [r["pred0"], r["pred_dca1"], r["pred_dca2"], r["pred1"], r["pred2"]], [r["pred_dca3"], pred4]
        return pred4, r["prob_volume2"].squeeze(1)


def __getattr__(name):
    # the reference also defines `hourglass` (gwcnet_dca_g.py:69-106); it lives in models/gwcnet.py here
    if name == "hourglass":
        from .gwcnet import hourglass
        return hourglass
    raise AttributeError(name)


def GwcNet_G(d):
    """factory expected by the reference's models/__init__.py:4-7 (present in gwcnet_dca{0,1,2}_g.py)"""
    return GwcNet(d, use_concat_volume=False)


def GwcNet_GC(d):
    return GwcNet(d, use_concat_volume=True)

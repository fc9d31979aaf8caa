"""Mirror of the reference's models/submodule.py for the symbols the DCANet path uses.

Hot-path functions (`build_gwc_volume`, `build_concat_volume`, `groupwise_correlation`,
`disparity_regression`) run as HIP kernels (dcanet_amd.ops); 3D conv blocks are parameter holders whose
arithmetic is done by dcanet_amd.ops.convbn3d.  The 2D networks (feature extractor, guidance, convex
up-sampler) are the callers either side of the path and stay on PyTorch-ROCm."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ._bootstrap import ensure as _ensure

ops = _ensure().ops


# ------------------------------------------------------------------ hot path: volumes + regression
def disparity_regression(x, maxdisp):
    """reference models/submodule.py:127-131"""
    assert len(x.shape) == 4
    assert x.shape[1] == maxdisp
    return ops.regression(x)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """reference models/submodule.py:134-145"""
    return ops.concat_volume(refimg_fea, targetimg_fea, maxdisp)


def groupwise_correlation(fea1, fea2, num_groups):
    """reference models/submodule.py:148-154 (= the disparity-0 plane of the gwc volume)."""
    B, C, H, W = fea1.shape
    assert C % num_groups == 0
    cost = ops.gwc_volume(fea1, fea2, 1, num_groups)[:, :, 0]
    assert cost.shape == (B, num_groups, H, W)
    return cost


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """reference models/submodule.py:157-167"""
    B, C, H, W = refimg_fea.shape
    assert C % num_groups == 0
    return ops.gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups)


# ------------------------------------------------------------------ 3D conv blocks (parameter holders)
def convbn_3d(in_channels, out_channels, kernel_size, stride, pad):
    """reference models/submodule.py:121-124: Sequential(Conv3d(bias=False), BatchNorm3d); same keys.
    Calling the returned module runs the HIP conv + BN (no activation)."""
    return ConvBn3d(in_channels, out_channels, kernel_size, stride, pad)


class ConvBn3d(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size, stride, pad):
        super().__init__(nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad,
                                   bias=False),
                         nn.BatchNorm3d(out_channels))
        assert (kernel_size, pad) in ((3, 1), (1, 0)) and stride in (1, 2)

    def forward(self, x, slope=1.0, res_pre=None, res_post=None, x2=None, alias=False, pack_out=False):
        """alias=True: returns (z, x') -- x' = x for the other consumers of x (ops._Conv3d.forward);
        pack_out=True: the result feeds exactly one 3x3x3 stride-1 ConvBn3d (ops.convbn3d)"""
        return ops.convbn3d(x, self[0], self[1], slope, res_pre, res_post, x2, alias, pack_out)


class ConvBnReLU3d(nn.Sequential):
    """Sequential(convbn_3d(...), ReLU): children '0' (ConvBn3d) and '1' (ReLU) like the reference."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, pad):
        super().__init__(ConvBn3d(in_channels, out_channels, kernel_size, stride, pad), nn.ReLU(inplace=True))

    def forward(self, x):
        return self[0](x, slope=0.0)


def conv3d_plain(x, conv):
    """nn.Conv3d(32, 1, 3, padding=1, bias=False) heads (classifiers)."""
    return ops.conv3d(x, conv.weight, conv.stride[0], False)


# ------------------------------------------------------------------ 2D neighbours of the path (PyTorch)
def convbn(in_channels, out_channels, kernel_size, stride, pad, dilation):
    """reference models/submodule.py:115-118"""
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                   padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_channels))


class BasicBlock(nn.Module):
    """reference models/submodule.py:251-273 (no ReLU after the residual add)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        if self.downsample is not None:
            x = self.downsample(x)
        return out + x


class BasicConv(nn.Module):
    """reference models/submodule.py:276-302 (2D, non-transposed use only on this path)."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, bn=True, relu=True, **kwargs):
        super().__init__()
        assert not deconv and not is_3d
        self.relu = relu
        self.use_bn = bn
        self.conv = nn.Conv2d(in_channels, out_channels, bias=False, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        x = self.conv(x)
        if self.use_bn:
            x = self.bn(x)
        return F.relu(x) if self.relu else x


class ResidualBlock(nn.Module):
    """reference models/submodule.py:305-355 with norm_fn='batch' (the only one Guidance uses)."""

    def __init__(self, in_planes, planes, norm_fn="batch", stride=1):
        super().__init__()
        assert norm_fn == "batch"
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = nn.BatchNorm2d(planes)
        self.norm2 = nn.BatchNorm2d(planes)
        if stride == 1:
            self.downsample = None
        else:
            self.norm3 = nn.BatchNorm2d(planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class Guidance(nn.Module):
    """reference models/submodule.py:395-460; returns {'g': (B,64,H/4,W/4)}."""

    def __init__(self, output_dim=64, norm_fn="batch"):
        super().__init__()
        assert norm_fn == "batch"
        self.norm_fn = norm_fn
        self.norm1 = nn.BatchNorm2d(32)
        self.conv_start = nn.Sequential(nn.Conv2d(3, 32, kernel_size=7, stride=2, padding=3), self.norm1,
                                        nn.ReLU(inplace=True))
        self.in_planes = 32
        self.layer1 = self._make_layer(32, stride=1)
        self.layer2 = self._make_layer(64, stride=2)
        self.conv_g0 = nn.Sequential(BasicConv(64, 64, kernel_size=3, padding=1),
                                     BasicConv(64, 64, kernel_size=3, padding=1))
        self.guidance = nn.Conv2d(64, output_dim, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layers = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                  ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.layer2(self.layer1(self.conv_start(x)))
        return {"g": self.guidance(self.conv_g0(x))}


class PropgationNet_4x(nn.Module):
    """reference models/submodule.py:357-373 (identical copy models/gwcnet_dca_g.py:108-124):
    9-neighbour convex x4 up-sampling of the 1/4-res disparity, values scaled by 4.  The mask conv stays on
    PyTorch-ROCm (2D); everything after it is one fused kernel."""

    def __init__(self, base_channels):
        super().__init__()
        self.base_channels = base_channels
        self.conv = nn.Sequential(convbn(base_channels, base_channels * 2, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                  nn.Conv2d(base_channels * 2, 9 * 16, kernel_size=(3, 3), stride=(1, 1), padding=1,
                                            dilation=(1, 1), bias=False))

    def forward(self, guidance, disp):
        # unfold(4*disp) . softmax_9(mask) . sum . pixel-shuffle as one HIP kernel (csrc/heads2d.hip)
        return ops.convex_upsample4(self.conv(guidance), disp)

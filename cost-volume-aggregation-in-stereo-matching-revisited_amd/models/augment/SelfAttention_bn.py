"""Mirror of the reference's models/augment/SelfAttention_bn.py (SelfAttentionBlock) on HIP kernels."""
import torch.nn as nn

from .._bootstrap import ensure as _ensure

ops = _ensure().ops


class _ProjLayer(nn.Sequential):
    """Sequential(Conv3d 1x1x1 (bias=False), BatchNorm3d, LeakyReLU(0.1)) -- keys '0.weight', '1.*'."""

    def __init__(self, cin, cout):
        super().__init__(nn.Conv3d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False), nn.BatchNorm3d(cout),
                         nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x):
        return ops.convbn3d(x, self[0], self[1], slope=0.1)


class _ProjStack(nn.Sequential):
    def forward(self, x):
        for layer in self:
            x = layer(x)
        return x


class SelfAttentionBlock(nn.Module):
    """reference SelfAttention_bn.py:14-98,136-160: 4 heads x 8 channels, attention along the disparity
    bins of each pixel; q/k use two projection layers, v/out one."""

    def __init__(self, key_in_channels, query_in_channels, transform_channels, out_channels, share_key_query,
                 query_downsample, key_downsample, key_query_num_convs, value_out_num_convs, key_query_norm,
                 value_out_norm, matmul_norm, with_out_project, **kwargs):
        super().__init__()
        if (query_downsample is not None or key_downsample is not None or not key_query_norm or not value_out_norm
                or not matmul_norm):
            raise NotImplementedError("only the configuration used by SemanticLevelContext is built")
        self.key_project = self.buildproject(key_in_channels, transform_channels, key_query_num_convs, True)
        if share_key_query:
            assert key_in_channels == query_in_channels
            self.query_project = self.key_project
        else:
            self.query_project = self.buildproject(query_in_channels, transform_channels, key_query_num_convs, True)
        self.value_project = self.buildproject(key_in_channels,
                                               transform_channels if with_out_project else out_channels,
                                               value_out_num_convs, True)
        self.out_project = None
        if with_out_project:
            self.out_project = self.buildproject(transform_channels, out_channels, value_out_num_convs, True)
        self.query_downsample = query_downsample
        self.key_downsample = key_downsample
        self.matmul_norm = matmul_norm
        self.transform_channels = transform_channels

    def forward(self, query_feats, key_feats):
        q = self.query_project(query_feats)
        k = self.key_project(key_feats)
        v = self.value_project(key_feats)
        ctx = ops.disparity_attention(q, k, v)      # head_dim = 8, scale 8**-0.5 (SelfAttention_bn.py:64,84-86)
        if self.out_project is not None:
            ctx = self.out_project(ctx)
        return ctx

    def buildproject(self, in_channels, out_channels, num_convs, use_norm):
        assert use_norm
        convs = [_ProjLayer(in_channels, out_channels)]
        for _ in range(num_convs - 1):
            convs.append(_ProjLayer(out_channels, out_channels))
        if len(convs) > 1:
            return _ProjStack(*convs)
        return convs[0]

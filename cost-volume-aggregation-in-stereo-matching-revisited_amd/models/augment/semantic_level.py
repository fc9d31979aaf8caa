"""Mirror of the reference's models/augment/semantic_level.py (SemanticLevelContext) on HIP kernels."""
import torch.nn as nn

from .._bootstrap import ensure as _ensure
from .SelfAttention_bn import SelfAttentionBlock

ops = _ensure().ops


class SemanticLevelContext(nn.Module):
    """reference semantic_level.py:14-128.  forward(x, preds): homogeneous-region context injection
    (closed form of the per-class loop, no host syncs) followed by the disparity cross-attention."""

    def __init__(self, feats_channels, transform_channels, reduction=8, concat_input=True, **kwargs):
        super().__init__()
        self.cross_attention = SelfAttentionBlock(
            key_in_channels=feats_channels, query_in_channels=feats_channels, transform_channels=transform_channels,
            out_channels=feats_channels, share_key_query=False, query_downsample=None, key_downsample=None,
            key_query_num_convs=2, value_out_num_convs=1, key_query_norm=True, value_out_norm=True,
            matmul_norm=True, with_out_project=True)

    def forward(self, x, preds):
        key_feats, _ = ops.context_inject(x, preds)          # feats_sl + inputs (semantic_level.py:102-126)
        return self.cross_attention(x, key_feats)

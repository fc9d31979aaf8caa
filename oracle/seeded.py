"""Tag-seeded test tensors (TEST INFRASTRUCTURE ONLY, see oracle/dcanet_oracle.py header).

`seeded_tensor(tag, shape)` is bit-identical on every machine with the same torch CPU
generator, so golden fixtures store only outputs plus a fingerprint of the inputs.
"""
import zlib

import torch


def seeded_tensor(tag: str, shape, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(tag.encode()) ^ 0x5EED) & 0x7FFFFFFF)
    return torch.randn(tuple(int(s) for s in shape), generator=g, dtype=torch.float32).to(dtype)


def thin(t):
    """Sub-sample large tensors (stride 4 on dim 0, repeatedly) so fixtures stay small; tests apply
    the same rule to what they compute."""
    while t.numel() > 16384 and t.shape[0] >= 4:
        t = t[::4]
    return t

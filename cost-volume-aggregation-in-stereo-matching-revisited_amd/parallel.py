"""Batch data-parallel helper for the DCANet training step: one process per GPU, the batch sharded on
dim 0, ONE collective per step -- an all-reduce (sum, then / world) of a flat fp32 gradient bucket over
RCCL/xGMI (backend "nccl" on ROCm) -- then the optimizer runs locally and identically on every rank.

MI355X-first replacement of the reference's `nn.DataParallel` (main_dca.py:54, train_kitti.py:74): no
per-step parameter broadcast, no scatter/gather through device 0.  BatchNorm statistics stay per replica,
exactly as plain nn.BatchNorm3d under DataParallel behaves in the reference."""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); returns (rank, local, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # DCA_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (tests / 1-GPU boxes)
            backend = os.environ.get("DCA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatGradBucket:
    """One contiguous fp32 buffer for all gradients (18.3 MB for GwcNet-G: one all-reduce message per step).

    Per step:  `zero()` drops the old .grad tensors so autograd ASSIGNS fresh gradients (no `grad += g` kernel per
    parameter); `all_reduce_mean()` gathers them into the flat buffer with one batched copy, all-reduces it, and
    re-points every parameter's .grad at its slice of the buffer (views, no scatter copies)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.zero()

    def zero(self):
        for p in self.params:
            p.grad = None

    def gather(self):
        """copies the freshly assigned .grad tensors into the flat buffer (parameters without a gradient -> 0).

        Note: a parameter that received no gradient this step gets a ZERO slice and `.grad` pointing at it (every rank
        must hand the collective the same buffer layout).  An optimizer therefore still updates it through its moments /
        weight decay, where the reference's `zero_grad()` + `step()` skips parameters whose `.grad` is None.  In DCANet's
        training step every parameter of the hot path receives a gradient, so the two coincide; pass only the parameters
        a step really trains if that matters."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def reduce_flat(self):
        """the collective alone (the flat buffer must already hold this rank's gradients: see gather())"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def all_reduce_mean(self):
        self.gather()
        self.reduce_flat()


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """dim-0 shard of a global batch (B must divide by world)."""
    assert t.shape[0] % world == 0, "global batch must be divisible by the number of ranks"
    per = t.shape[0] // world
    return t[rank * per:(rank + 1) * per]

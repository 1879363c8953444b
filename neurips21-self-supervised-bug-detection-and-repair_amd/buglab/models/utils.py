"""Counterpart of reference buglab/models/utils.py for the HIP path: segment ops (:15-48), the
optimiser factory (:51-52) and the warm-up scheduler (:55-66).  `compute_generator_loss`
(:101-179, selector training) is a SURVEY section 8f 'next' row."""
from typing import Optional

import torch

from buglab.models import hip_ops
from buglab.runtime.optim import FlatAdam
from buglab.runtime.trainer import AbstractScheduler


def scatter_log_softmax(src: torch.Tensor, index: torch.Tensor, dim: int = -1, eps: float = 1e-12) -> torch.Tensor:
    """reference :15-28.  Builds the CSR of `index` on the host (one small D2H); hot callers pass
    the collator's prebuilt CSR to `hip_ops.segment_log_softmax` directly instead."""
    from buglab.data.collate import segments_from_index

    if not torch.is_floating_point(src):
        raise ValueError("`scatter_log_softmax` can only be computed over tensors with floating point data types.")
    idx = index.detach().cpu().numpy()
    n = int(idx.max()) + 1 if idx.size else 0
    ptr, items = segments_from_index(idx, n)
    return hip_ops.segment_log_softmax(src.float(), torch.from_numpy(ptr).to(src.device), torch.from_numpy(items).to(src.device), n, eps)


def optimizer(p, lr: float = 0.0001) -> FlatAdam:
    """reference :51-52 (`torch.optim.Adam(p, lr)`), as the fused flat-buffer Adam."""
    return FlatAdam(p, lr=lr, clip_gradient_norm=0.0, num_warmup_steps=0)


class LinearWarmupScheduler(AbstractScheduler):
    """reference :55-66.  The linear warm-up factor is folded into FlatAdam's fused step; this
    object only configures it and keeps the per-step `step()` call site working."""

    def __init__(self, optimizer: FlatAdam, num_warmup_steps: int = 800, last_epoch=-1):
        optimizer.warmup = num_warmup_steps

    def step(self, epoch_idx: int, epoch_step: int) -> None:
        pass

"""Counterpart of reference buglab/models/utils.py for the HIP path: segment ops (:15-48), the
optimiser factory (:51-52), the warm-up scheduler (:55-66) and the selector ("generator") loss
`compute_generator_loss` (:101-179)."""
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


def compute_generator_loss(arg_swap_logprobs, arrange, candidate_rewrite_idxs, candidate_symbol_to_location_group,
                           localization_logprobs, loss_type, pair_rewrite_idxs, rewrite_logprobs, rewrite_to_graph_id,
                           rewrite_to_location_group, swapped_pair_to_call_location_group, text_repair_logprobs,
                           text_rewrite_idxs, varmisuse_logprobs, gen_group_ptr=None, gen_group_items=None,
                           gen_num_groups: Optional[int] = None):
    """reference utils.py:101-179, same positional arguments.  The reference compacts the observed
    entries with `masked_select` and runs torch_scatter ops on the compacted vectors; here the
    segment kernels run on the FULL-length vectors through a CSR that lists only the observed
    positions per graph (built by the collator from the host-side `rewrite_logprobs`, or here with
    one small D2H if a caller does not pass it), so nothing is compacted or re-indexed on the device."""
    dev = rewrite_logprobs.device
    B = arrange.shape[0]
    L = lambda t: t.long()
    gen = torch.cat([torch.zeros_like(rewrite_logprobs[: rewrite_logprobs.shape[0] - B]), localization_logprobs[-B:]])  # :117-121
    gen = gen.index_add(0, L(text_rewrite_idxs), localization_logprobs[L(rewrite_to_location_group)] + text_repair_logprobs)  # :123-125
    gen = gen.index_add(0, L(candidate_rewrite_idxs), localization_logprobs[L(candidate_symbol_to_location_group)] + varmisuse_logprobs)
    gen = gen.index_add(0, L(pair_rewrite_idxs), localization_logprobs[L(swapped_pair_to_call_location_group)] + arg_swap_logprobs)
    if gen_group_ptr is None:
        import numpy as np

        from buglab.data.collate import _csr

        lp = rewrite_logprobs.detach().cpu().numpy()
        index = np.concatenate([rewrite_to_graph_id.cpu().numpy().astype(np.int64), np.arange(B, dtype=np.int64)])
        observed = np.flatnonzero(~np.isinf(lp))
        gen_num_groups = int(index[observed].max()) + 1 if observed.size else 0
        ptr, order = _csr(index[observed], gen_num_groups)
        gen_group_ptr = torch.from_numpy(ptr).to(dev)
        gen_group_items = torch.from_numpy(observed[order].astype(np.int32)).to(dev)
    ng = int(gen_num_groups)
    sel = L(gen_group_items)
    det = rewrite_logprobs
    if loss_type in ("norm-kl", "norm-rmse", "classify-max-loss"):
        gen_n = hip_ops.segment_log_softmax(gen, gen_group_ptr, gen_group_items, ng)  # :143-146
        if loss_type == "norm-rmse":
            det_n = hip_ops.segment_log_softmax(det.contiguous(), gen_group_ptr, gen_group_items, ng)
            return (torch.logaddexp(det_n[sel], gen_n[sel]) ** 2).mean()  # :148-152
        if loss_type == "norm-kl":
            failed = torch.log(torch.clamp(1.0 - det.exp(), min=1e-30))  # :154-159 (unobserved: log(1 - 0) = 0, never read)
            renorm = hip_ops.segment_log_softmax(failed, gen_group_ptr, gen_group_items, ng)
            kl = failed[sel].exp() * (renorm[sel] - gen_n[sel])  # :163-165
            return kl.sum() / ng  # scatter_sum(...).mean() over the ng groups
        # classify-max-loss (:167-169): arg-min of the detection log-prob per graph (first minimum)
        _, arg, _, _, _ = hip_ops.segment_max((-det).unsqueeze(1).contiguous(), gen_group_ptr, gen_group_items, ng)
        return -gen_n[L(arg[:, 0])].mean()
    if loss_type == "expectation":
        return (gen[sel].exp() * det[sel]).sum() / ng  # :172-176
    raise ValueError(f"Unknown loss type `{loss_type}`")

"""Bug-localization head (reference buglab/models/layers/localizationmodule.py), HIP-backed.

Same module/method names and maths; differences in mechanism only:
  * the three nn.Linear layers and the per-graph max pool run in libbuglab_hip (gathered GEMM,
    segmented max), and `[candidate ; pooled[graph]]` is never materialised;
  * the segmented log-softmax works on a CSR of the ids (built by the collator) instead of
    torch_scatter atomics;
  * metric counters stay on the device: the reference's four `.cpu()` syncs per step
    (localizationmodule.py:108-113) happen once, when `_module_metrics()` is read.
Weights are stored [in, out] (the reference's nn.Linear stores [out, in]).
"""
import math
from typing import Any, Callable, Dict

import torch
from torch import nn

from buglab.models import hip_ops
from buglab.runtime.module import ModuleWithMetrics


class LocalizationModule(ModuleWithMetrics):
    def __init__(self, representation_size: int, buggy_samples_weight_schedule: Callable[[int], float],
                 abstain_weight: float = 0.0):
        super().__init__()
        H = representation_size
        u = lambda shape, fan_in: nn.Parameter(torch.empty(shape).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in)))
        self.Ws, self.bs = u((H, H), H), u((H,), H)  # _summary_repr       (:22)
        self.W1, self.b1 = u((2 * H, H), 2 * H), u((H,), 2 * H)  # _l1     (:23)
        self.w = u((H,), H)  # _repr_to_localization_score, bias=False     (:24)
        self._buggy_samples_weight_schedule = buggy_samples_weight_schedule
        self._abstain_weight = abstain_weight
        self._epoch_idx = 0
        self._stats = None
        self._stats_source = None  # the owning module's fused loss assembly keeps the counters (hip_ops.bug_loss): () -> [5] or None

    def _all_stats(self):
        ext = self._stats_source() if self._stats_source is not None else None
        if ext is None:
            return self._stats
        return ext if self._stats is None else self._stats + ext

    def _reset_module_metrics(self) -> None:
        st = self._all_stats()
        if st is not None and self.training and float(st[0]) > 0:
            self._epoch_idx += 1  # "Assumes that module metrics are reset once per epoch" (:32-36)
        self._stats = None

    def _module_metrics(self) -> Dict[str, Any]:
        st = self._all_stats()
        if st is None:
            return {}
        total, correct, no_bug, no_bug_correct, loss = (float(x) for x in st.tolist())
        if total == 0:
            return {}
        return {
            "Localization Accuracy": correct / total,
            "No Bug Recall": no_bug_correct / no_bug if no_bug > 0 else float("nan"),
            "Localization Loss": loss / total,
            "Weight of Buggy Samples": self._buggy_samples_weight_schedule(self._epoch_idx),
        }

    def compute_localization_scores(self, node_reprs, candidate_nodes, candidate_to_sample_idx, num_samples, candidate_ptr):
        """reference :56-60: the candidates' scores before the NO_BUG logit and the log-softmax (one C call)."""
        return hip_ops.localization_scores(node_reprs, candidate_nodes, candidate_to_sample_idx, candidate_ptr, num_samples,
                                           self.Ws, self.bs, self.W1, self.b1, self.w)

    def compute_localization_logprobs(self, node_reprs, candidate_nodes, candidate_to_sample_idx, num_samples,
                                      candidate_ptr, loc_group_ptr, loc_group_items):
        """reference :54-79.  Takes the node-state matrix plus the candidate row ids (the gather
        `reprs[candidate_nodes]` of gnn.py:170-172 is folded into the GEMM operand load).
        -> (ids [C+B] int32, logprobs [C+B], arange [B])"""
        dev = node_reprs.device
        C = candidate_nodes.shape[0]
        # :56-60 in one call: summary GEMM -> per-graph max pool -> [candidate ; pooled] GEMM + sigmoid -> score
        scores = hip_ops.localization_scores(node_reprs, candidate_nodes, candidate_to_sample_idx, candidate_ptr, num_samples,
                                             self.Ws, self.bs, self.W1, self.b1, self.w)
        arange = torch.arange(num_samples, dtype=torch.int32, device=dev)
        scores_with_no_bug = torch.cat((scores, torch.ones(num_samples, dtype=torch.float32, device=dev)))  # :63-68
        ids = torch.cat((candidate_to_sample_idx, arange))
        logprobs = hip_ops.segment_log_softmax(scores_with_no_bug, loc_group_ptr, loc_group_items, num_samples)  # :72-77
        return ids, logprobs, arange

    def forward(self, node_reprs, candidate_nodes, candidate_to_sample_idx, has_bug, correct_candidate_idxs,
                candidate_ptr, loc_group_ptr, loc_group_items):
        B = has_bug.shape[0]
        C = candidate_nodes.shape[0]
        ids, log_probs, arange = self.compute_localization_logprobs(
            node_reprs, candidate_nodes, candidate_to_sample_idx, B, candidate_ptr, loc_group_ptr, loc_group_items)
        correct = torch.where(has_bug, correct_candidate_idxs.long(), arange.long() + C)  # :86-90
        lp = log_probs[correct].clamp(max=math.log(0.995))  # :92-93
        if self._abstain_weight > 0:  # :95-100
            lp = lp + torch.where(has_bug, self._abstain_weight * log_probs[arange.long() + C], torch.zeros_like(lp))
        with torch.no_grad():  # :102-114, without host syncs
            # argmax per graph: candidates of graph b are rows candidate_ptr[b]..candidate_ptr[b+1], NO_BUG is row C+b
            seg_max, seg_arg = hip_ops.segment_max(log_probs[:C].unsqueeze(1).contiguous(), candidate_ptr, None, B)[:2]
            no_bug_lp = log_probs[C:]
            pred = torch.where((seg_arg[:, 0] >= 0) & (seg_max[:, 0] >= no_bug_lp), seg_arg[:, 0].long(), arange.long() + C)
            ok = pred == correct
            nb = has_bug.logical_not()
            # (torch.full is a fill kernel; torch.tensor(B, device=...) would be a blocking copy from pageable memory)
            stats = torch.stack([torch.full((), float(B), device=lp.device), ok.sum().float(), nb.sum().float(),
                                 (nb & ok).sum().float(), -lp.sum()])
            self._stats = stats if self._stats is None else self._stats + stats
        w_buggy = self._buggy_samples_weight_schedule(self._epoch_idx)
        if w_buggy == 1.0:
            return -lp.mean()  # :116-117
        weights = torch.where(has_bug, torch.full_like(lp, w_buggy), torch.ones_like(lp))  # :119-123
        return -(lp * weights).sum() / weights.sum()

"""Repair scorers (reference buglab/models/layers/fixermodules.py + mlp.py), HIP-backed.

Each is MLP(k*H -> H -> 1) with ReLU (mlp.py:6-20) over a concatenation of gathered node states;
the concatenation is consumed directly by the gathered GEMM, the H -> 1 layer is a row-dot kernel.
`CandidatePairSelectorModule` uses the representation width of its input instead of the
reference's never-assigned `self._input_dim` (fixermodules.py:120) -- documented deviation.
Metric counters stay on the device (no per-step `int(tensor)` syncs, cf. :49-50, 94-95, 143-144).
"""
import math
from typing import Any, Dict

import torch
from torch import nn

from buglab.models import hip_ops
from buglab.runtime.module import ModuleWithMetrics


class _ScorerBase(ModuleWithMetrics):
    _metric_name = ""
    _stats_name = ""

    def __init__(self, input_dim: int, hidden: int):
        super().__init__()
        b = 1.0 / math.sqrt(input_dim)
        self.W1 = nn.Parameter(torch.empty(input_dim, hidden).uniform_(-b, b))
        self.b1 = nn.Parameter(torch.empty(hidden).uniform_(-b, b))
        b2 = 1.0 / math.sqrt(hidden)
        self.w2 = nn.Parameter(torch.empty(hidden).uniform_(-b2, b2))
        self.b2 = nn.Parameter(torch.empty(1).uniform_(-b2, b2))
        self._counts = None
        self._counts_source = None  # the owning module's fused loss assembly keeps the counters: () -> [2] (hits, total) or None

    def _score(self, sources):
        return hip_ops.mlp_score(sources, self.W1, self.b1, self.w2, self.b2)  # one C call forward, one backward

    def _reset_module_metrics(self) -> None:
        self._counts = None

    def _module_metrics(self) -> Dict[str, Any]:
        ext = self._counts_source() if self._counts_source is not None else None
        if self._counts is None and ext is None:
            return {}
        correct, total = (int(a) + int(b) for a, b in zip(self._counts.tolist() if self._counts is not None else (0, 0),
                                                           ext.tolist() if ext is not None else (0, 0)))
        if total == 0:
            return {}
        return {self._metric_name: correct / total, self._stats_name: f"{correct / total:.2%} ({correct}/{total})"}

    def forward(self, logprobs, targets, selected_fixes=None):
        """-logprob of the correct candidates (reference :41-53, 86-98, 134-147)."""
        targets = targets.long()
        if selected_fixes is not None:
            with torch.no_grad():
                c = torch.stack([selected_fixes[targets].sum().long(),  # (full: a fill kernel; torch.tensor would be a blocking copy)
                                 torch.full((), int(targets.shape[0]), dtype=torch.long, device=targets.device)])
                self._counts = c if self._counts is None else self._counts + c
        return -logprobs[targets]


class TextRepairModule(_ScorerBase):
    _metric_name, _stats_name = "Text Repair Fixer Accuracy", "Text Repair Fixer Stats"

    def __init__(self, input_representation_size: int, rewrite_vocab_size: int):
        super().__init__(2 * input_representation_size, input_representation_size)
        self.emb = nn.Parameter(torch.randn(rewrite_vocab_size, input_representation_size))  # nn.Embedding init

    def compute_rewrite_logits(self, node_reprs, target_rewrite_nodes, candidate_rewrites):
        """reference :31-39: MLP([Emb[rewrite] ; h[node]])."""
        return self._score([(self.emb, candidate_rewrites), (node_reprs, target_rewrite_nodes)])


class SingleCandidateNodeSelectorModule(_ScorerBase):
    _metric_name, _stats_name = "VarMisuse Repair Fixer Accuracy", "VarMisuse Repair Fixer Stats"

    def __init__(self, input_representation_size: int):
        super().__init__(2 * input_representation_size, input_representation_size)

    def compute_per_slot_log_probability(self, node_reprs, slot_nodes, candidate_nodes):
        """reference :65-73: MLP([h[slot] ; h[candidate]])."""
        return self._score([(node_reprs, slot_nodes), (node_reprs, candidate_nodes)])


class CandidatePairSelectorModule(_ScorerBase):
    _metric_name, _stats_name = "ArgSwap Repair Fixer Accuracy", "ArgSwap Repair Fixes Stats"

    def __init__(self, input_node_representation: int):
        super().__init__(3 * input_node_representation, input_node_representation)

    def compute_per_pair_logits(self, node_reprs, call_nodes, pair_a_nodes, pair_b_nodes):
        """reference :110-124: MLP([h[call] ; h[a] ; h[b]])."""
        return self._score([(node_reprs, call_nodes), (node_reprs, pair_a_nodes), (node_reprs, pair_b_nodes)])

"""GNN bug detector / repair model -- MI355X-native counterpart of reference buglab/models/gnn.py.

`GnnBugLabModule` (device side) keeps the reference's method names and call contract
(`forward(**minibatch) -> loss`, `compute_localization_logprobs(graph_data)`,
`_compute_repair_logprobs(gnn_output, ...)`, metric hooks); `GnnBugLabModel` (host side) keeps
`update_metadata_from / build_neural_module / tensorize / initialize_minibatch /
extend_minibatch_with / finalize_minibatch / predict`.  Underneath, the typed message passing and
all scoring heads run in libbuglab_hip (buglab.models.hip_ops) and minibatches are built by the
NumPy CSR collator (buglab.data.collate).
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from buglab.data import collate as C
from buglab.data.collate import BaseTensorizedBugLabGnn, TensorizedGraphData
from buglab.models import hip_ops
from buglab.models.layers.fixermodules import CandidatePairSelectorModule, SingleCandidateNodeSelectorModule, TextRepairModule
from buglab.models.layers.localizationmodule import LocalizationModule
from buglab.models.layers.messagepassing import GnnOutput, GraphNeuralNetwork
from buglab.runtime.module import ModuleWithMetrics

LOGGER = logging.getLogger(__name__)


class _FusedStatsSlice:
    """Picklable "read my slice of the owner's fused-loss counters" callable handed to the sub-modules."""

    def __init__(self, owner, lo: int, hi: int):
        self.owner, self.lo, self.hi = owner, lo, hi

    def __call__(self):
        st = self.owner._fused_stats
        return None if st is None else st[self.lo:self.hi]


class GnnBugLabModule(ModuleWithMetrics):
    """reference gnn.py:55-322."""

    def __init__(self, gnn: GraphNeuralNetwork, rewrite_vocabulary_size: int, use_all_gnn_layer_outputs: bool = False,
                 generator_loss_type: Optional[str] = "norm-kl",
                 buggy_samples_weight_schedule: Callable[[int], float] = lambda _: 1.0, dropout_base_seed: int = 0):
        super().__init__()
        self._generator_loss_type = generator_loss_type
        self._gnn = gnn
        self._use_all_gnn_layer_outputs = use_all_gnn_layer_outputs
        if use_all_gnn_layer_outputs:  # reference :68-74
            in_f = gnn.input_node_state_dim + sum(l.output_state_dimension for l in gnn.message_passing_layers)
            b = 1.0 / np.sqrt(in_f)
            self.summarization_W = nn.Parameter(torch.empty(in_f, gnn.output_node_state_dim).uniform_(-b, b))
            self.summarization_b = nn.Parameter(torch.empty(gnn.output_node_state_dim).uniform_(-b, b))
        H = gnn.output_node_state_dim
        self._localization_module = LocalizationModule(H, buggy_samples_weight_schedule=buggy_samples_weight_schedule)
        self._buggy_samples_weight_schedule = buggy_samples_weight_schedule
        self._text_repair_module = TextRepairModule(H, rewrite_vocabulary_size)
        self._varmisuse_module = SingleCandidateNodeSelectorModule(H)
        self._argswap_module = CandidatePairSelectorModule(H)
        self._epoch_idx = 0
        self._acc = None
        # counters of the fused loss assembly (hip_ops.bug_loss: one kernel for everything between the logits and the loss);
        # the sub-modules read their slices when metrics are reported
        self._fused_stats = None
        self._localization_module._stats_source = _FusedStatsSlice(self, 0, 5)
        self._text_repair_module._counts_source = _FusedStatsSlice(self, 5, 7)
        self._varmisuse_module._counts_source = _FusedStatsSlice(self, 7, 9)
        self._argswap_module._counts_source = _FusedStatsSlice(self, 9, 11)
        self._dropout_base_seed = int(dropout_base_seed)
        self._dropout_step = 0

    @property
    def use_all_gnn_layer_outputs(self):
        return self._use_all_gnn_layer_outputs

    @property
    def gnn(self):
        return self._gnn

    # ---- metrics (reference :95-114) ----------------------------------------------------------
    def _all_acc(self):
        ext = None if self._fused_stats is None else self._fused_stats[11:15]
        if ext is None:
            return self._acc
        return ext if self._acc is None else self._acc + ext

    def _reset_module_metrics(self) -> None:
        acc = self._all_acc()
        if acc is not None and self.training and float(acc[3]) > 0:
            self._epoch_idx += 1
        self._acc = None  # [loss sum, repair loss sum, total buggy samples, num batches]
        # nn.Module.modules() yields this module BEFORE its children: let the localization head see its counters (it advances
        # its own epoch index on reset) before they are dropped; its own turn then finds nothing and does nothing
        self._localization_module._reset_module_metrics()
        self._fused_stats = None

    def _module_metrics(self) -> Dict[str, Any]:
        acc = self._all_acc()
        if acc is None:
            return {}
        loss, repair, samples, batches = (float(x) for x in acc.tolist())
        m = {}
        if samples > 0:
            m["Repair Loss"] = repair / samples
        if batches > 0:
            m["Loss"] = loss / batches
        return m

    # ---- forward pieces -------------------------------------------------------------------------
    def _next_dropout_seed(self) -> Optional[int]:
        if not self.training:
            return None
        self._dropout_step += 1
        return (self._dropout_base_seed * 0x9E3779B1 + self._dropout_step) & 0xFFFFFFFF

    def _compute_gnn_output(self, graph_data, dropout_seed=None) -> GnnOutput:
        out: GnnOutput = self._gnn(**graph_data, return_all_states=self._use_all_gnn_layer_outputs, dropout_seed=dropout_seed)
        if self._use_all_gnn_layer_outputs:  # reference :118-121
            out = out._replace(output_node_representations=hip_ops.gather_linear(
                [(out.output_node_representations, None)], self.summarization_W, self.summarization_b, "none"))
        idx_all = graph_data.get("head_gather_idx")
        if idx_all is not None and idx_all.shape[0] > 0:
            # all scorers read node rows through ONE gather: backward then has a single [N, H] scatter
            # instead of a zero-filled [N, H] buffer (and an accumulation) per gathered operand
            local, spans = graph_data["head_local_idx"], graph_data["head_spans"]
            out = out._replace(head_node_representations=hip_ops.gather_rows(out.output_node_representations, idx_all),
                               head_idx_references={k: local[s : s + n] for k, (s, n) in spans.items()})
        return out

    def overlap_parameter_groups(self):
        """Parameter lists, one per message-passing layer in forward order: the units whose gradient all-reduce a
        data-parallel run starts as soon as that layer's backward has been launched (runtime/optim.py)."""
        from buglab.models.layers.messagepassing import MlpMessagePassingLayer

        return [list(l.parameters()) for l in self._gnn.message_passing_layers if isinstance(l, MlpMessagePassingLayer)]

    @staticmethod
    def _head_inputs(gnn_output: GnnOutput):
        if gnn_output.head_node_representations is not None:
            return gnn_output.head_node_representations, gnn_output.head_idx_references
        return gnn_output.output_node_representations, gnn_output.node_idx_references

    def compute_localization_logprobs(self, graph_data: Dict[str, Any], dropout_seed=None):
        """reference :125-142."""
        gnn_output = self._compute_gnn_output(graph_data, dropout_seed)
        h, refs = self._head_inputs(gnn_output)
        ids, logprobs, arange = self._localization_module.compute_localization_logprobs(
            h,
            refs["candidate_nodes"],
            gnn_output.node_graph_idx_reference["candidate_nodes"],
            gnn_output.num_graphs,
            graph_data["candidate_ptr"], graph_data["loc_group_ptr"], graph_data["loc_group_items"])
        return ids, logprobs, gnn_output, arange

    def _repair_logits(self, gnn_output: GnnOutput, target_rewrites):
        """The three scorers' logits (reference :261-293), each a [0] tensor when its head has no candidates."""
        h, refs = self._head_inputs(gnn_output)
        zeros = lambda: torch.zeros(0, dtype=torch.float32, device=h.device)
        text = (self._text_repair_module.compute_rewrite_logits(h, refs["target_rewrite_nodes"], target_rewrites)
                if target_rewrites.shape[0] > 0 else zeros())
        var = (self._varmisuse_module.compute_per_slot_log_probability(h, refs["varmisused_node_ids"], refs["candidate_symbol_node_ids"])
               if refs["varmisused_node_ids"].shape[0] > 0 else zeros())
        swap = (self._argswap_module.compute_per_pair_logits(h, refs["call_node_ids"], refs["candidate_swapped_a"], refs["candidate_swapped_b"])
                if refs["call_node_ids"].shape[0] > 0 else zeros())
        return text, var, swap

    def _fused_loss_applicable(self) -> bool:
        # one buggy-sample weight for the localization and the repair part (the two schedules are the same object in every
        # constructor of this repository; they could only differ if someone re-wired a sub-module by hand)
        loc = self._localization_module
        return loc._buggy_samples_weight_schedule(loc._epoch_idx) == self._buggy_samples_weight_schedule(self._epoch_idx)

    def _forward_fused_loss(self, gnn_output, graph_data, has_bug, correct_candidate_node_idxs, target_rewrites,
                            rewrite_to_location_group, candidate_symbol_to_location_group, swapped_pair_to_call_location_group,
                            correct_rewrite_idxs, correct_candidate_symbols, correct_swapped_pair, repair_group_ptr, repair_group_items):
        """reference :221-251 with localizationmodule.py:63-124 and fixermodules' forward()s folded in (hip_ops.bug_loss)."""
        h, refs = self._head_inputs(gnn_output)
        B = has_bug.shape[0]
        scores = self._localization_module.compute_localization_scores(
            h, refs["candidate_nodes"], gnn_output.node_graph_idx_reference["candidate_nodes"], B, graph_data["candidate_ptr"])
        text, var, swap = self._repair_logits(gnn_output, target_rewrites)
        logits = torch.cat((text, var, swap))  # :295
        ix = hip_ops.BugLossIndex(graph_data["loc_group_ptr"], graph_data["loc_group_items"], graph_data["candidate_ptr"], has_bug,
                                  correct_candidate_node_idxs, repair_group_ptr, repair_group_items,
                                  (rewrite_to_location_group, candidate_symbol_to_location_group, swapped_pair_to_call_location_group),
                                  (correct_rewrite_idxs, correct_candidate_symbols, correct_swapped_pair),
                                  int(repair_group_ptr.shape[0]) - 1)
        loss, stats = hip_ops.bug_loss(scores, logits, (text.shape[0], var.shape[0], swap.shape[0]), ix,
                                       self._buggy_samples_weight_schedule(self._epoch_idx), self._localization_module._abstain_weight)
        with torch.no_grad():  # :244-249 and the sub-modules' counters, without host syncs
            self._fused_stats = stats if self._fused_stats is None else self._fused_stats + stats
        return loss

    def _compute_repair_logprobs(self, gnn_output: GnnOutput, target_rewrites, rewrite_to_location_group,
                                 candidate_symbol_to_location_group, swapped_pair_to_call_location_group,
                                 repair_group_ptr=None, repair_group_items=None, num_repair_groups: Optional[int] = None):
        """reference :253-322.  One log-softmax over the location groups of ALL three scorers'
        logits; the CSR of the concatenated group ids comes from the collator (or is built here
        from the three id vectors when a caller does not supply it)."""
        h, refs = self._head_inputs(gnn_output)
        dev = h.device
        zeros = lambda: torch.zeros(0, dtype=torch.float32, device=dev)
        if target_rewrites.shape[0] > 0:
            text_logits = self._text_repair_module.compute_rewrite_logits(h, refs["target_rewrite_nodes"], target_rewrites)
        else:
            text_logits = zeros()
        if refs["varmisused_node_ids"].shape[0] > 0:
            var_logits = self._varmisuse_module.compute_per_slot_log_probability(
                h, refs["varmisused_node_ids"], refs["candidate_symbol_node_ids"])
        else:
            var_logits = zeros()
        if refs["call_node_ids"].shape[0] > 0:
            swap_logits = self._argswap_module.compute_per_pair_logits(
                h, refs["call_node_ids"], refs["candidate_swapped_a"], refs["candidate_swapped_b"])
        else:
            swap_logits = zeros()
        all_logits = torch.cat((text_logits, var_logits, swap_logits))  # :295
        logit_groups = torch.cat((rewrite_to_location_group, candidate_symbol_to_location_group,
                                  swapped_pair_to_call_location_group))  # :296-298
        if repair_group_ptr is None:
            g = logit_groups.cpu().numpy()
            n = int(num_repair_groups if num_repair_groups is not None else (g.max() + 1 if g.size else 0))
            ptr, items = C.segments_from_index(g, n)
            repair_group_ptr, repair_group_items = torch.from_numpy(ptr).to(dev), torch.from_numpy(items).to(dev)
        nseg = repair_group_ptr.shape[0] - 1
        logprobs = hip_ops.segment_log_softmax(all_logits, repair_group_ptr, repair_group_items, nseg)  # :299
        sizes = [text_logits.shape[0], var_logits.shape[0], swap_logits.shape[0]]
        text_lp, var_lp, swap_lp = torch.split(logprobs, sizes)
        with torch.no_grad():  # :304-311  is this candidate the arg-max of its location group?
            # log-softmax is monotone inside a group and the group's maximum has the largest logprob:
            # x == max_g  <=>  x - max_g == 0  <=>  logprob == -log(sum_g exp(x - max_g) + eps) = max logprob
            if all_logits.numel() > 0:
                gmax = torch.full((nseg,), -float("inf"), device=dev).scatter_reduce(0, logit_groups.long(), all_logits, "amax")
                sel = gmax[logit_groups.long()] == all_logits
            else:
                sel = torch.zeros(0, dtype=torch.bool, device=dev)
            text_sel, var_sel, swap_sel = torch.split(sel, sizes)
        return swap_lp, text_lp, var_lp, (swap_sel, text_sel, var_sel)

    def forward(self, *, graph_data: Dict[str, Any], correct_candidate_node_idxs, has_bug: torch.Tensor,
                target_rewrites, rewrite_to_location_group, correct_rewrite_idxs, text_rewrite_idxs,
                candidate_symbol_to_location_group, correct_candidate_symbols, candidate_rewrite_idxs,
                swapped_pair_to_call_location_group, correct_swapped_pair, pair_rewrite_idxs, rewrite_to_graph_id,
                rewrite_logprobs: Optional[torch.Tensor] = None, repair_group_ptr=None, repair_group_items=None,
                num_repair_groups=None, gen_group_ptr=None, gen_group_items=None, gen_num_groups=None,
                dropout_seed: Optional[int] = None, **kwargs):
        """reference :144-251 (keyword-only arguments, visualisation extras ignored)."""
        if dropout_seed is None:
            dropout_seed = self._next_dropout_seed()
        gnn_output = self._compute_gnn_output(graph_data, dropout_seed)
        if rewrite_logprobs is None and hip_ops.FUSED_LOSS and repair_group_ptr is not None and self._fused_loss_applicable():
            # detector training: everything between the scorers' logits and the loss in one kernel per direction
            return self._forward_fused_loss(gnn_output, graph_data, has_bug, correct_candidate_node_idxs, target_rewrites,
                                            rewrite_to_location_group, candidate_symbol_to_location_group,
                                            swapped_pair_to_call_location_group, correct_rewrite_idxs, correct_candidate_symbols,
                                            correct_swapped_pair, repair_group_ptr, repair_group_items)
        swap_lp, text_lp, var_lp, (swap_sel, text_sel, var_sel) = self._compute_repair_logprobs(
            gnn_output, target_rewrites, rewrite_to_location_group, candidate_symbol_to_location_group,
            swapped_pair_to_call_location_group, repair_group_ptr, repair_group_items, num_repair_groups)
        if rewrite_logprobs is not None:  # selector / generator branch, reference :189-219
            from buglab.models.utils import compute_generator_loss

            h, refs = self._head_inputs(gnn_output)
            _, loc_lp, arange = self._localization_module.compute_localization_logprobs(
                h, refs["candidate_nodes"],
                gnn_output.node_graph_idx_reference["candidate_nodes"], has_bug.shape[0],
                graph_data["candidate_ptr"], graph_data["loc_group_ptr"], graph_data["loc_group_items"])
            loss = compute_generator_loss(
                swap_lp, arange, candidate_rewrite_idxs, candidate_symbol_to_location_group, loc_lp, self._generator_loss_type,
                pair_rewrite_idxs, rewrite_logprobs, rewrite_to_graph_id, rewrite_to_location_group,
                swapped_pair_to_call_location_group, text_lp, text_rewrite_idxs, var_lp, gen_group_ptr, gen_group_items, gen_num_groups)
            with torch.no_grad():  # :214-217
                acc = torch.stack([loss.detach(), torch.zeros((), device=loss.device), torch.zeros((), device=loss.device),
                                   torch.ones((), device=loss.device)])
                self._acc = acc if self._acc is None else self._acc + acc
            return loss
        h, refs = self._head_inputs(gnn_output)
        loc_loss = self._localization_module(
            h,
            refs["candidate_nodes"],
            gnn_output.node_graph_idx_reference["candidate_nodes"],
            has_bug, correct_candidate_node_idxs,
            graph_data["candidate_ptr"], graph_data["loc_group_ptr"], graph_data["loc_group_items"])
        text_loss = self._text_repair_module(text_lp, correct_rewrite_idxs, selected_fixes=text_sel)
        var_loss = self._varmisuse_module(var_lp, correct_candidate_symbols, selected_fixes=var_sel)
        swap_loss = self._argswap_module(swap_lp, correct_swapped_pair, selected_fixes=swap_sel)
        w_buggy = self._buggy_samples_weight_schedule(self._epoch_idx)
        repair_loss = (text_loss.sum() + var_loss.sum() + swap_loss.sum()) * w_buggy  # :240-242
        B = has_bug.shape[0]
        loss = loc_loss + repair_loss / B  # :251
        with torch.no_grad():  # :244-249 without the five host syncs
            acc = torch.stack([loss.detach(), repair_loss.detach(), has_bug.sum().float(), torch.ones((), device=loss.device)])
            self._acc = acc if self._acc is None else self._acc + acc
        return loss


def const_weight_schedule(_epoch_idx: int, weight: float = 1.0) -> float:
    return weight


def build_gnn_mlp_module(hidden_state_size: int = 128, num_layers: int = 8, num_edge_types: int = 16,
                         vocabulary_size: int = 15000, max_num_subtokens: int = 6, rewrite_vocabulary_size: int = 48,
                         dropout_rate: float = 0.2, message_activation: str = "gelu",
                         buggy_samples_weight: float = 1.0, dropout_base_seed: int = 0, model: str = "gnn-mlp",
                         edge_feature_size: int = 0, edge_vocabulary_size: int = 0,
                         embedder_dropout_rate: Optional[float] = None,
                         message_activation_placement: str = "aggregated",
                         embedder_dropout_placement: str = "after_pooling",
                         message_aggregation_function: str = "max") -> GnnBugLabModule:
    """Device module for given hyper-parameters without a metadata pass (bench / tests / synthetic
    runs).  `GnnBugLabModel.build_neural_module()` goes through the same constructors.
    `embedder_dropout_rate`: dropout of the node embedder; None = `dropout_rate` (the oracle's single-rate
    configuration the parity tests use).  The registry-built gnn models pass 0.0 unless `node_representations` carries a
    rate: the reference's `gnn()` does not hand `dropout_rate` to the node embedder (modelregistry.py:79-82)."""
    from functools import partial

    from buglab.models.gnnlayerdefs import create_mlp_mp_layers
    from buglab.models.layers.messagepassing import SubtokenEmbedder, TokenEmbedder

    embed = SubtokenEmbedder(vocabulary_size, hidden_state_size, max_num_subtokens,
                             dropout_rate if embedder_dropout_rate is None else embedder_dropout_rate, embedder_dropout_placement)
    edge_embed = TokenEmbedder(edge_vocabulary_size, edge_feature_size) if edge_feature_size > 0 else None
    if model == "ggnn":
        assert edge_embed is None
        from buglab.models.gnnlayerdefs import create_ggnn_mp_layers

        recipe = create_ggnn_mp_layers(hidden_state_size, dropout_rate, num_edge_types)
    else:
        recipe = create_mlp_mp_layers(hidden_state_size, dropout_rate, num_edge_types, features_dimension=edge_feature_size,
                                      num_layers=num_layers, message_activation=message_activation,
                                      message_activation_placement=message_activation_placement,
                                      message_aggregation_function=message_aggregation_function)
    return GnnBugLabModule(GraphNeuralNetwork(embed, recipe, edge_embedder=edge_embed), rewrite_vocabulary_size,
                           buggy_samples_weight_schedule=partial(const_weight_schedule, weight=buggy_samples_weight),
                           dropout_base_seed=dropout_base_seed)


# =================================================================================================
# host side
from buglab.models.basemodel import AbstractBugLabModel  # noqa: E402
from buglab.models.graphmodel import GraphNeuralNetworkModel  # noqa: E402
from buglab.representations.data import BugLabData  # noqa: E402
from buglab.runtime.neuralmodel import AbstractNeuralModel  # noqa: E402


class GnnBugLabModel(AbstractNeuralModel, AbstractBugLabModel):
    """reference gnn.py:325-645."""

    def __init__(self, gnn_model: GraphNeuralNetworkModel, use_all_gnn_layer_outputs: bool = False,
                 generator_loss_type: Optional[str] = "classify-max-loss",
                 buggy_samples_weight_schedule: Callable[[int], float] = None):
        super().__init__()
        self._init()
        self._gnn_model = gnn_model
        self._use_all_gnn_layer_outputs = use_all_gnn_layer_outputs
        self._generator_loss_type = generator_loss_type
        from functools import partial

        self._buggy_samples_weight_schedule = buggy_samples_weight_schedule or partial(const_weight_schedule, weight=1.0)

    @property
    def gnn_model(self):
        return self._gnn_model

    @property
    def use_all_gnn_layer_outputs(self):
        return self._use_all_gnn_layer_outputs

    def update_metadata_from(self, datapoint: BugLabData) -> None:
        graph_data, _ = BugLabData.as_graph_data(datapoint)
        self._gnn_model.update_metadata_from(graph_data)

    def finalize_metadata(self) -> None:
        self._gnn_model.finalize_metadata()

    def build_neural_module(self) -> GnnBugLabModule:
        return GnnBugLabModule(self._gnn_model.build_neural_module(),
                               rewrite_vocabulary_size=len(self._target_rewrite_ops),
                               use_all_gnn_layer_outputs=self._use_all_gnn_layer_outputs,
                               generator_loss_type=self._generator_loss_type,
                               buggy_samples_weight_schedule=self._buggy_samples_weight_schedule)

    def tensorize(self, datapoint: BugLabData) -> Optional[BaseTensorizedBugLabGnn]:
        """reference :361-429."""
        graph_data, target_location_node_idx = BugLabData.as_graph_data(datapoint)
        if "candidate_rewrite_logprobs" in datapoint and datapoint["candidate_rewrite_logprobs"] is not None:
            assert not self._tensorize_only_at_target_location_rewrites
        (target_rewrite_node_ids, target_rewrites, target_rewrite_to_location_group, correct_rewrite_target,
         text_rewrite_original_idx, varmisused_node_ids, candidate_symbol_to_varmisused_location, candidate_symbol_node_ids,
         correct_candidate_symbol_node, varmisuse_rewrite_original_idx, call_node_ids, candidate_swapped_node_ids,
         correct_swapped_pair, swapped_pair_to_call, swapped_rewrite_original_ids, repr_location_group_ids,
         ) = self._compute_rewrite_data(datapoint, graph_data.reference_nodes["candidate_nodes"])
        refs = graph_data.reference_nodes
        refs["target_rewrite_nodes"] = target_rewrite_node_ids
        refs["varmisused_node_ids"] = varmisused_node_ids
        refs["candidate_symbol_node_ids"] = candidate_symbol_node_ids
        refs["call_node_ids"] = call_node_ids
        refs["candidate_swapped_node_ids"] = (np.asarray(candidate_swapped_node_ids, dtype=np.int32).reshape(-1, 2)
                                              if len(candidate_swapped_node_ids) else np.zeros((0, 2), dtype=np.int32))
        # `is not None` where the reference tests truthiness (:399-401), so index 0 also counts
        assert sum(x is not None for x in (correct_rewrite_target, correct_candidate_symbol_node, correct_swapped_pair)) <= 1, \
            "No more than one node should be correct."
        tensorized_graph = self._gnn_model.tensorize(graph_data)
        if tensorized_graph is None:
            return None
        return BaseTensorizedBugLabGnn(
            graph_data=tensorized_graph, target_location_node_idx=target_location_node_idx,
            target_rewrites=target_rewrites, target_rewrite_to_location_group=target_rewrite_to_location_group,
            correct_rewrite_target=correct_rewrite_target, text_rewrite_original_idx=text_rewrite_original_idx,
            candidate_symbol_to_varmisused_node=candidate_symbol_to_varmisused_location,
            correct_candidate_symbol_node=correct_candidate_symbol_node,
            candidate_rewrite_original_idx=varmisuse_rewrite_original_idx,
            swapped_pair_to_call=swapped_pair_to_call, correct_swapped_pair=correct_swapped_pair,
            pair_rewrite_original_idx=swapped_rewrite_original_ids,
            num_rewrite_locations_considered=len(repr_location_group_ids),
            rewrite_logprobs=datapoint.get("candidate_rewrite_logprobs", None))

    # minibatching: the reference appends element by element (:431-542) and builds ~15 tensors from
    # Python lists (:544-604); here samples are only collected and `finalize_minibatch` does one
    # vectorised collate + one H2D copy (buglab.data.collate).
    def initialize_minibatch(self) -> Dict[str, Any]:
        return {"samples": [], "num_nodes": 0}

    def extend_minibatch_with(self, tensorized_datapoint: BaseTensorizedBugLabGnn, partial_minibatch: Dict[str, Any]) -> bool:
        partial_minibatch["samples"].append(tensorized_datapoint)
        partial_minibatch["num_nodes"] += tensorized_datapoint.graph_data.num_nodes
        return partial_minibatch["num_nodes"] < self._gnn_model.stop_extending_minibatch_after_num_nodes

    def collate_minibatch(self, accumulated_minibatch_data: Dict[str, Any]) -> Dict[str, Any]:
        """The host half of `finalize_minibatch`: NumPy arrays only, so loader processes can run it
        (runtime/shardloader.py) and ship whole minibatches."""
        return C.collate_samples(accumulated_minibatch_data["samples"], self._gnn_model.num_presented_edge_types)

    def finalize_minibatch(self, accumulated_minibatch_data: Dict[str, Any], device: Union[str, torch.device]) -> Dict[str, Any]:
        return C.to_device(self.collate_minibatch(accumulated_minibatch_data), device)

    def predict(self, data: Iterator[BugLabData], trained_nn: GnnBugLabModule, device, parallelize: bool
                ) -> Iterator[Tuple[BugLabData, Dict[int, float], List[float]]]:
        """reference :606-645."""
        trained_nn.eval()
        with torch.no_grad(), self._tensorize_all_location_rewrites():
            for mb_data, original_datapoints in self.minibatch_iterator(
                    self.tensorize_dataset(data, return_input_data=True, parallelize=parallelize), device,
                    max_minibatch_size=50, parallelize=parallelize):
                ids, loc_lp, gnn_output, _ = trained_nn.compute_localization_logprobs(mb_data["graph_data"])
                swap_lp, text_lp, var_lp, _ = trained_nn._compute_repair_logprobs(
                    gnn_output, mb_data["target_rewrites"], mb_data["rewrite_to_location_group"],
                    mb_data["candidate_symbol_to_location_group"], mb_data["swapped_pair_to_call_location_group"],
                    mb_data["repair_group_ptr"], mb_data["repair_group_items"])
                yield from self._iter_per_sample_results(mb_data, ids, loc_lp, swap_lp, gnn_output.num_graphs,
                                                         original_datapoints, text_lp, var_lp)

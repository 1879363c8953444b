"""Sequence bug detector / repair models (`seq-great`, `seq-rat`) -- MI355X counterpart of reference
buglab/models/seqmodel.py.

`SeqBugLabModel` (host side) keeps the reference's constructor kwargs and the AbstractNeuralModel methods
(`update_metadata_from / finalize_metadata / build_neural_module / tensorize / initialize_minibatch /
extend_minibatch_with / finalize_minibatch / predict`).  `SeqBugLabModule` (device side) is the reference's
module with the same head wiring: token embedding -> + learned positional table -> LayerNorm -> dropout ->
relational transformer layers -> localization / repair heads over token positions (seqmodel.py:351-396).

Mechanism: a padded minibatch [B, L] is laid out as B * L "nodes" (node b * L + i = token i of sample b), so the
scoring heads, the loss assembly, the selector loss and the un-batching are the ones the graph model already runs
on the HIP path (`GnnBugLabModule`): the sequence encoder only replaces the message-passing stack.  The encoder's
kernels: subtoken embedder (csrc/bl_graph_ops.hip), relational transformer block
(buglab/models/layers/relational_transformer.py -> csrc/bl_seq_ops.hip + the MFMA GEMMs).

`seq-transformer` (torch.nn.TransformerEncoderLayer) and `seq-gru` (nn.GRU) are not on the HIP path.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, Iterator, List, NamedTuple, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from buglab.data import collate as C
from buglab.data.collate import BaseTensorizedBugLabGnn, TensorizedGraphData
from buglab.data.seqcollate import edge_csr
from buglab.models import hip_ops
from buglab.models.basemodel import AbstractBugLabModel
from buglab.models.gnn import GnnBugLabModule, const_weight_schedule
from buglab.models.graphmodel import StrElementRepresentationModel
from buglab.models.hip_ops import Dropout, RelEdges
from buglab.models.layers.messagepassing import GnnOutput, SubtokenEmbedder
from buglab.models.layers.relational_transformer import RelationalTransformerEncoderLayer
from buglab.models.layers.gru import BiGRULayer
from buglab.models.layers.transformer import TransformerEncoderLayer
from buglab.representations.tokenseq import project_graph_to_tokens
from buglab.runtime.neuralmodel import AbstractNeuralModel

LOGGER = logging.getLogger(__name__)
MAX_POSITIONS = 5000  # rows of the learned positional table (reference seqmodel.py:86)


class SequenceEncoder(nn.Module):
    """Token embedder + positional table + input LayerNorm + relational transformer stack
    (`_compute_output_representation`, reference seqmodel.py:351-396), behind the call contract
    `GnnBugLabModule` uses for its node encoder: `encoder(**graph_data, return_all_states, dropout_seed) -> GnnOutput`."""

    def __init__(self, token_embedder: SubtokenEmbedder, embedding_dim: int, num_edge_types: int, num_layers: int, num_heads: int,
                 intermediate_dimension: int, dropout_rate: float, layer_type: str = "great", rezero_mode: str = "off",
                 normalisation_mode: str = "postnorm"):
        super().__init__()
        if layer_type not in ("great", "rat", "transformer", "gru"):
            raise ValueError(f"Unrecognized layer type `{layer_type}`.")  # reference seqmodel.py:128
        D = embedding_dim
        self.embed = token_embedder
        self.positional_encoding = nn.Parameter(torch.randn(1, MAX_POSITIONS, D))
        self.input_norm_g, self.input_norm_b = nn.Parameter(torch.ones(D)), nn.Parameter(torch.zeros(D))
        self.dropout_rate = dropout_rate
        self.layer_type = layer_type
        # `transformer`: torch.nn.TransformerEncoderLayer's arithmetic (reference seqmodel.py:108-118); edges are not looked at
        # `gru`: torch.nn.GRU(D, D // 2, num_layers, bidirectional) as a stack of bidirectional layers (reference seqmodel.py:119-126)
        self.layers = nn.ModuleList([BiGRULayer(D, D // 2) for _ in range(num_layers)]) if layer_type == "gru" else nn.ModuleList([
            TransformerEncoderLayer(d_model=D, nhead=num_heads, dim_feedforward=intermediate_dimension, dropout=dropout_rate)
            for _ in range(num_layers)]) if layer_type == "transformer" else nn.ModuleList([
            RelationalTransformerEncoderLayer(d_model=D, key_query_dimension=D // num_heads, value_dimension=D // num_heads,
                                              nhead=num_heads, num_edge_types=max(1, num_edge_types), dim_feedforward=intermediate_dimension,
                                              dropout=dropout_rate, use_edge_value_biases=layer_type == "rat", rezero_mode=rezero_mode,
                                              normalisation_mode=normalisation_mode)
            for _ in range(num_layers)])
        for l in self.layers:
            l.output_state_dimension = D
        self.input_node_state_dim = D
        self.output_node_state_dim = D

    @property
    def message_passing_layers(self):
        return list(self.layers)

    def forward(self, *, token_ids, token_lens, seq_lens, seq_batch, seq_len, erow_ptr, ekey, ecode, node_to_graph,
                reference_node_ids, reference_node_graph_idx, num_graphs, return_all_states: bool = False,
                dropout_seed: Optional[int] = None, tok_occ=None, tok_chunk_ptr=None, tok_chunk_id=None, **_unused) -> GnnOutput:
        B, L = int(seq_batch), int(seq_len)
        training = self.training and dropout_seed is not None
        seed = int(dropout_seed or 0)
        mk = lambda stream: Dropout(self.dropout_rate if training else 0.0, seed, stream)
        tok_csr = (tok_occ, tok_chunk_ptr, tok_chunk_id) if tok_occ is not None else None
        emb = self.embed(token_ids, token_lens, mk(0), tok_csr)  # [B * L, D]
        if getattr(self, "layer_type", "great") == "gru":
            x = emb  # reference seqmodel.py:363: positions, input LayerNorm and dropout are for the transformer variants only
        else:
            pos = self.positional_encoding[0, :L].unsqueeze(0).expand(B, L, -1).reshape(B * L, -1)
            x = hip_ops.add_layernorm(emb, pos, self.input_norm_g, self.input_norm_b)  # LayerNorm(embedding + position)
            x = hip_ops.dropout_rows(x, mk(1))
        valid = (torch.arange(L, device=x.device, dtype=torch.int32)[None, :] < seq_lens[:, None]).reshape(B * L, 1)
        x = x * valid  # `output_representation *= token_mask` (:372)
        h0 = x
        edges = RelEdges(erow_ptr, ekey, ecode, int(ekey.shape[0]))
        states = [h0]
        chain = {}  # (packed activations from one fused layer call to the next: hip_ops.great_layer)
        for i, layer in enumerate(self.layers):
            x = layer(x, seq_lens, edges, B, L, dropout_seed=seed if training else None, dropout_stream=8 * (i + 1), chain=chain)
            states.append(x)
        out = torch.cat(states, dim=-1) if return_all_states else x
        return GnnOutput(h0, out, node_to_graph, reference_node_ids, reference_node_graph_idx, int(num_graphs))


class SeqBugLabModule(GnnBugLabModule):
    """reference seqmodel.py:65-396 (`SeqBugLabModule`).  Heads, loss assembly, selector loss and metrics are
    `GnnBugLabModule`'s (the reference's two modules share that code line for line: seqmodel.py:164-349 vs
    gnn.py:144-322); `forward(**minibatch)` takes the minibatch `SeqBugLabModel.finalize_minibatch` builds."""

    def __init__(self, encoder: SequenceEncoder, rewrite_vocabulary_size: int,
                 buggy_samples_weight_schedule: Callable[[int], float] = lambda _: 1.0, generator_loss_type: Optional[str] = "norm-kl"):
        super().__init__(encoder, rewrite_vocabulary_size, use_all_gnn_layer_outputs=False, generator_loss_type=generator_loss_type,
                         buggy_samples_weight_schedule=buggy_samples_weight_schedule)

    def _compute_output_representation(self, graph_data) -> torch.Tensor:
        """[B * L, D] token representations (reference :351-396 returns them as [B, L, D])."""
        return self._compute_gnn_output(graph_data).output_node_representations


class SeqTensorizedSample(NamedTuple):
    """A tensorised sample: the same fields as the graph model's, over token positions, plus the node -> token map."""

    base: BaseTensorizedBugLabGnn
    node_mappings: Dict[int, int]
    edge_kinds: Tuple[str, ...]  # kinds present, aligned with base.graph_data.adjacency_lists


class SeqBugLabModel(AbstractNeuralModel, AbstractBugLabModel):
    """reference seqmodel.py:399-1028."""

    def __init__(self, representation_size: int, max_subtoken_vocab_size: int, dropout_rate: float, layer_type: str = "great",
                 max_seq_size: int = 500, num_heads: int = 8, num_layers: int = 6, intermediate_dimension_size: int = 2048,
                 buggy_samples_weight_schedule: Callable[[int], float] = None, generator_loss_type: Optional[str] = "classify-max-loss",
                 rezero_mode: str = "off", normalisation_mode: str = "postnorm"):
        super().__init__()
        self._init()
        from functools import partial

        self._edge_kinds_seen = set()
        self.edge_types: Optional[List[str]] = None
        self._dropout_rate, self._representation_size, self._layer_type = dropout_rate, representation_size, layer_type
        self._max_seq_size = max_seq_size
        self._token_embedder = StrElementRepresentationModel(token_splitting="subtoken", embedding_size=representation_size,
                                                             dropout_rate=dropout_rate, vocabulary_size=max_subtoken_vocab_size,
                                                             subtoken_combination="max")
        self._buggy_samples_weight_schedule = buggy_samples_weight_schedule or partial(const_weight_schedule, weight=1.0)
        self._num_heads, self._num_layers = num_heads, num_layers
        self._intermediate_dimension_size = intermediate_dimension_size
        self._generator_loss_type = generator_loss_type
        self._rezero_mode, self._normalisation_mode = rezero_mode, normalisation_mode

    @property
    def token_embedder(self):
        return self._token_embedder

    # ---- metadata ---------------------------------------------------------------------------------
    def _token_data(self, graph):
        try:
            return project_graph_to_tokens(graph)
        except Exception as ex:  # the reference logs and drops the sample (:619-626, :637-641)
            LOGGER.debug("Error in generating token sequence for %s: %r", graph.get("path"), ex)
            return None

    def update_metadata_from(self, datapoint) -> None:
        td = self._token_data(datapoint["graph"])
        if td is None:
            return
        labels, _, edges, _ = td
        for n in labels:
            self._token_embedder.update_metadata_from(n)
        self._edge_kinds_seen.update(edges.keys())

    def finalize_metadata(self) -> None:
        self._token_embedder.finalize_metadata()
        # the reference keeps `list(set)` (:631), i.e. an order that depends on the process' string hashing; sorted here
        self.edge_types = sorted(self._edge_kinds_seen)
        self._edge_kinds_seen = None

    def build_neural_module(self) -> SeqBugLabModule:
        enc = SequenceEncoder(self._token_embedder.build_neural_module(), self._token_embedder.embedding_size, len(self.edge_types),
                              self._num_layers, self._num_heads, self._intermediate_dimension_size, self._dropout_rate,
                              layer_type=self._layer_type, rezero_mode=self._rezero_mode, normalisation_mode=self._normalisation_mode)
        return SeqBugLabModule(enc, rewrite_vocabulary_size=len(self._target_rewrite_ops),
                               buggy_samples_weight_schedule=self._buggy_samples_weight_schedule,
                               generator_loss_type=self._generator_loss_type)

    # ---- tensorize (reference :633-720) -------------------------------------------------------------
    def tensorize(self, datapoint) -> Optional[SeqTensorizedSample]:
        if "candidate_rewrite_logprobs" in datapoint and datapoint["candidate_rewrite_logprobs"] is not None:
            assert not self._tensorize_only_at_target_location_rewrites
        td = self._token_data(datapoint["graph"])
        if td is None:
            return None
        labels, node_to_pos, edges, _reference_positions = td
        if len(labels) > self._max_seq_size:
            return None
        # the mapped candidate positions may contain duplicates: the graph -> token map is not injective (:650-652)
        candidate_node_idxs, inv = np.unique(datapoint["graph"]["reference_nodes"], return_inverse=True)
        candidate_positions = np.array([node_to_pos[int(n)] for n in candidate_node_idxs], dtype=np.int32)
        target = datapoint["target_fix_action_idx"]
        target_node_idx = int(inv[target]) if target is not None else None
        (target_rewrite_node_ids, target_rewrites, target_rewrite_to_location_group, correct_rewrite_target, text_rewrite_original_idx,
         varmisused_node_ids, candidate_symbol_to_varmisused_location, candidate_symbol_node_ids, correct_candidate_symbol_node,
         varmisuse_rewrite_original_idx, call_node_ids, candidate_swapped_node_ids, correct_swapped_pair, swapped_pair_to_call,
         swapped_rewrite_original_ids, repr_location_group_ids) = self._compute_rewrite_data(datapoint, candidate_node_idxs)
        m = lambda ids: np.array([node_to_pos[int(n)] for n in ids], dtype=np.int32)
        ids, lens = self._token_embedder.tensorize_nodes(labels)
        kinds = tuple(k for k in self.edge_types if k in edges) if self.edge_types is not None else tuple(sorted(edges))
        if self.edge_types is not None:
            unseen = sorted(k for k in edges if k not in self.edge_types and len(edges[k]))
            if unseen:  # the reference's edge-type vocabulary lookup raises a KeyError here (seqmodel.py:776-777)
                raise KeyError(f"edge kinds {unseen} were not seen when the metadata was computed")
        adj = [np.asarray(edges.get(k, ()), dtype=np.int32).reshape(-1, 2) for k in (self.edge_types if self.edge_types is not None else kinds)]
        refs = {
            "candidate_nodes": candidate_positions,
            "target_rewrite_nodes": m(target_rewrite_node_ids),
            "varmisused_node_ids": m(varmisused_node_ids),
            "candidate_symbol_node_ids": m(candidate_symbol_node_ids),
            "call_node_ids": m(call_node_ids),
            "candidate_swapped_node_ids": (np.array([(node_to_pos[int(a)], node_to_pos[int(b)]) for a, b in candidate_swapped_node_ids],
                                                    dtype=np.int32).reshape(-1, 2)),
        }
        base = BaseTensorizedBugLabGnn(
            graph_data=TensorizedGraphData(ids, lens, adj, refs), target_location_node_idx=target_node_idx,
            target_rewrites=target_rewrites, target_rewrite_to_location_group=target_rewrite_to_location_group,
            correct_rewrite_target=correct_rewrite_target, text_rewrite_original_idx=text_rewrite_original_idx,
            candidate_symbol_to_varmisused_node=candidate_symbol_to_varmisused_location,
            correct_candidate_symbol_node=correct_candidate_symbol_node, candidate_rewrite_original_idx=varmisuse_rewrite_original_idx,
            swapped_pair_to_call=swapped_pair_to_call, correct_swapped_pair=correct_swapped_pair,
            pair_rewrite_original_idx=swapped_rewrite_original_ids, num_rewrite_locations_considered=len(repr_location_group_ids),
            rewrite_logprobs=datapoint.get("candidate_rewrite_logprobs", None))
        return SeqTensorizedSample(base, dict(node_to_pos), kinds)

    # ---- minibatching (reference :722-975) ----------------------------------------------------------
    def initialize_minibatch(self) -> Dict[str, Any]:
        return {"samples": []}

    def extend_minibatch_with(self, tensorized_datapoint: SeqTensorizedSample, partial_minibatch: Dict[str, Any]) -> bool:
        partial_minibatch["samples"].append(tensorized_datapoint)
        return True  # the reference never stops extending a sequence minibatch early (:868)

    def collate_minibatch(self, accumulated_minibatch_data: Dict[str, Any]) -> Dict[str, Any]:
        return collate_sequences(accumulated_minibatch_data["samples"], len(self.edge_types))

    def finalize_minibatch(self, accumulated_minibatch_data: Dict[str, Any], device: Union[str, torch.device]) -> Dict[str, Any]:
        return C.to_device(self.collate_minibatch(accumulated_minibatch_data), device)

    def predict(self, data: Iterator, trained_nn: SeqBugLabModule, device, parallelize: bool
                ) -> Iterator[Tuple[Any, Dict[int, float], List[float]]]:
        """reference :977-1028."""
        trained_nn.eval()
        with torch.no_grad(), self._tensorize_all_location_rewrites():
            for mb_data, original_datapoints in self.minibatch_iterator(
                    self.tensorize_dataset(data, return_input_data=True, parallelize=parallelize), device,
                    max_minibatch_size=50, parallelize=parallelize):
                ids, loc_lp, enc_out, _ = trained_nn.compute_localization_logprobs(mb_data["graph_data"])
                swap_lp, text_lp, var_lp, _ = trained_nn._compute_repair_logprobs(
                    enc_out, mb_data["target_rewrites"], mb_data["rewrite_to_location_group"],
                    mb_data["candidate_symbol_to_location_group"], mb_data["swapped_pair_to_call_location_group"],
                    mb_data["repair_group_ptr"], mb_data["repair_group_items"])
                yield from self._iter_per_sample_results(mb_data, ids, loc_lp, swap_lp, enc_out.num_graphs, original_datapoints,
                                                         text_lp, var_lp, node_mappings=mb_data["node_mappings"])


def collate_sequences(samples: List[SeqTensorizedSample], num_edge_types: int) -> Dict[str, Any]:
    """B tensorised sequences -> one padded [B, L] minibatch (NumPy), L = the longest sequence rounded up to a
    multiple of 4.  Token i of sample b becomes "node" b * L + i of the graph collator's layout, so every index array
    of the heads (candidates, rewrite nodes, location groups, CSRs of the log-softmaxes) comes from the same code the
    graph model uses; the sequence-specific arrays are the per-sample lengths and the query-row CSR of the edges."""
    B = len(samples)
    L = max((s.base.graph_data.num_nodes for s in samples), default=1)
    L = max(4, (L + 3) // 4 * 4)
    padded = []
    for s in samples:
        g = s.base.graph_data
        n, S = g.token_ids.shape
        ids = np.zeros((L, S), dtype=np.int32)
        lens = np.ones(L, dtype=np.int32)  # padding: token id 0, one subtoken (reference :886-888)
        ids[:n], lens[:n] = g.token_ids, g.token_lens
        padded.append(s.base._replace(graph_data=TensorizedGraphData(ids, lens, g.adjacency_lists, g.reference_nodes)))
    mb = C.collate_samples(padded, num_edge_types)
    gd = mb["graph_data"]
    seq_lens = np.array([s.base.graph_data.num_nodes for s in samples], dtype=np.int32)
    types = np.repeat(np.arange(num_edge_types), np.diff(gd["type_ptr"]))
    e = np.stack([gd["msg_src"] // L, gd["msg_src"] % L, gd["msg_tgt"] % L], axis=1) if gd["msg_src"].size else np.zeros((0, 3), np.int64)
    gd["erow_ptr"], gd["ekey"], gd["ecode"] = edge_csr(e, types, B, L)
    gd["seq_lens"], gd["seq_batch"], gd["seq_len"] = seq_lens, B, L
    mb["node_mappings"] = [s.node_mappings for s in samples]
    return mb

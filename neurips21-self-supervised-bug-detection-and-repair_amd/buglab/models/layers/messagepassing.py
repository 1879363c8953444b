"""GNN device modules: node embedder, MlpMessagePassingLayer, ConcatResidualLayer and the
GraphNeuralNetwork container -- the roles ptgnn plays in the reference (call sites
buglab/models/gnnlayerdefs.py:5-39, modelregistry.py:59-90, gnn.py:70-76,116-123).

Arithmetic spec (frozen here because ptgnn's source is unavailable):
  message    m_e  = [h_src ; h_tgt] @ W[type(e)]              W: [T, 2*Din, Dm], no bias
  aggregate  a_v  = max over incoming messages (0 if none; ties -> lowest message id)
  activation message_activation_placement = "aggregated" (default, ptgnn's order as recollected): a_v <- gelu(a_v);
             "message" (this repository's rounds 1-5): m_e <- gelu(m_e) before the max
  update     h'_v = Dropout(tanh(LayerNorm(a_v) @ Wd + bd))
  edge features (features_dimension F > 0): m_e = [h_src ; h_tgt ; f_e] @ W[type(e)], W: [T, 2*Din + F, Dm],
             f_e = edge-embedding row of the edge's feature token (reversed edge: its forward edge's; self loop: pad)
Every FLOP runs in libbuglab_hip (buglab.models.hip_ops); this file only owns parameters.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, NamedTuple, Optional

import torch
from torch import nn

from buglab.models import hip_ops
from buglab.models.hip_ops import Dropout, GraphIndex


class GnnOutput(NamedTuple):
    """Field names follow ptgnn's GnnOutput as used at reference gnn.py:119-139,264-289,631."""

    input_node_representations: torch.Tensor
    output_node_representations: torch.Tensor
    node_to_graph_idx: torch.Tensor
    node_idx_references: Dict[str, torch.Tensor]
    node_graph_idx_reference: Dict[str, torch.Tensor]
    num_graphs: int
    # compact copy of the node rows the heads reference (one gather for all scorers) and the references
    # re-indexed into it; None -> the heads gather from output_node_representations themselves
    head_node_representations: Optional[torch.Tensor] = None
    head_idx_references: Optional[Dict[str, torch.Tensor]] = None


MAX_MESSAGE_DIMENSION = 512  # csrc/bl_graph_ops.hip: a wavefront of the per-node kernels holds one row of <= 8 x 64 channels


def _check_message_width(dm: int, who: str) -> None:
    """The segmented max / LayerNorm / node-gradient kernels keep a whole message row in one wavefront's registers (DISPATCH_NV,
    <= 512 channels).  gnn-mlp's concat layers have 2 x hidden message channels, so hidden_state_size <= 256 (every BASELINE
    configuration: 128 / 256).  Said here, at construction, rather than as a BL_EINVAL from the first forward pass."""
    if dm > MAX_MESSAGE_DIMENSION:
        raise NotImplementedError(f"{who}: message_dimension {dm} > {MAX_MESSAGE_DIMENSION}: the HIP per-node kernels hold a message row "
                                  f"of at most {MAX_MESSAGE_DIMENSION} channels per wavefront (gnn-mlp: hidden_state_size <= 256, its "
                                  "ConcatResidual layers carry 2 x hidden channels)")


def _uniform_(t: torch.Tensor, bound: float) -> torch.Tensor:
    return t.uniform_(-bound, bound)


class SubtokenEmbedder(nn.Module):
    """StrElementRepresentationModel in "subtoken" mode (modelregistry.py:59-82); subtoken_combination "max" (the registry's default),
    "sum" or "mean"."""

    def __init__(self, vocabulary_size: int, embedding_size: int, max_num_subtokens: int = 6, dropout_rate: float = 0.0,
                 dropout_placement: str = "after_pooling", subtoken_combination: str = "max"):
        super().__init__()
        if dropout_placement not in ("after_pooling", "before_pooling"):
            raise ValueError(f"dropout_placement must be 'after_pooling' or 'before_pooling' (got {dropout_placement!r})")
        if subtoken_combination not in hip_ops.POOLINGS:
            raise ValueError(f"subtoken_combination must be one of {hip_ops.POOLINGS} (got {subtoken_combination!r})")
        self.subtoken_combination = subtoken_combination
        self.embedding_size = embedding_size
        self.max_num_subtokens = max_num_subtokens
        self.dropout_rate = dropout_rate
        # where the embedder's dropout sits relative to the max over subtokens: on the pooled rows (default -- the last statement
        # of ptgnn's SubtokenUnitEmbedder.forward as recollected) or on the embedded subtokens before it
        self.dropout_placement = dropout_placement
        self.table = nn.Parameter(torch.randn(vocabulary_size, embedding_size))

    def forward(self, token_ids, token_lens, drop: Dropout, tok_csr=None):
        before = getattr(self, "dropout_placement", "after_pooling") == "before_pooling"  # (older pickles: after)
        return hip_ops.embed_subtoken_max(self.table, token_ids, token_lens, drop, tok_csr, dropout_before_pooling=before,
                                          combination=getattr(self, "subtoken_combination", "max"))  # (older pickles: max)


class TokenEmbedder(nn.Module):
    """StrElementRepresentationModel in "token" mode: the edge-feature embedding (modelregistry.py:70-74).  The table is read
    as a gathered source of every layer's message GEMM: forward here is the identity on the table."""

    def __init__(self, vocabulary_size: int, embedding_size: int):
        super().__init__()
        if embedding_size % 4 != 0:
            raise ValueError("edge_feature_size must be a multiple of 4 (row alignment of the gathered GEMM operand)")
        self.embedding_size = embedding_size
        self.table = nn.Parameter(torch.randn(vocabulary_size, embedding_size))  # nn.Embedding init


class MlpMessagePassingLayer(nn.Module):
    """kwargs as at the reference call site gnnlayerdefs.py:6-23."""

    def __init__(self, input_state_dimension: int, message_dimension: int, output_state_dimension: int,
                 num_edge_types: int, message_aggregation_function: str = "max", dropout_rate: float = 0.0,
                 features_dimension: int = 0, message_activation: str = "gelu",
                 message_activation_placement: str = "aggregated"):
        super().__init__()
        hip_ops.message_activation_code(message_activation, message_activation_placement)  # validates both
        # "max" is what the reference's recipe passes (gnnlayerdefs.py:11,21); ptgnn's "sum" / "mean" run in the one-call layer form
        # with the activation on the aggregate (or none) and without edge features
        if message_aggregation_function not in hip_ops.AGGREGATIONS:
            raise ValueError(f"message_aggregation_function must be one of {hip_ops.AGGREGATIONS} (got {message_aggregation_function!r})")
        if message_aggregation_function != "max" and (features_dimension > 0 or (message_activation == "gelu" and message_activation_placement == "message")):
            raise NotImplementedError("sum / mean aggregation: without edge features, message activation on the aggregate or none "
                                      "(a per-message activation would need the [E, Dm] messages again in backward)")
        self.message_aggregation_function = message_aggregation_function
        din, dm, dout, T = input_state_dimension, message_dimension, output_state_dimension, num_edge_types
        _check_message_width(dm, "MlpMessagePassingLayer")
        self.input_state_dimension, self.message_dimension, self.output_state_dimension = din, dm, dout
        self.num_edge_types, self.dropout_rate, self.message_activation = T, dropout_rate, message_activation
        self.message_activation_placement = message_activation_placement
        self.features_dimension = F = int(features_dimension)  # edge features: the message input is [h_src ; h_tgt ; f_e]
        self.W = nn.Parameter(_uniform_(torch.empty(T, 2 * din + F, dm), 1.0 / math.sqrt(2 * din + F)))
        self.ln_g = nn.Parameter(torch.ones(dm))
        self.ln_b = nn.Parameter(torch.zeros(dm))
        self.Wd = nn.Parameter(_uniform_(torch.empty(dm, dout), math.sqrt(6.0 / (dm + dout))))
        self.bd = nn.Parameter(_uniform_(torch.empty(dout), 1.0 / math.sqrt(dm)))

    def _msg_act(self) -> str:
        # a module pickled before round 6 has no placement attribute: it was trained with the per-message placement
        return hip_ops.message_activation_code(self.message_activation, getattr(self, "message_activation_placement", "message"))

    def forward(self, node_states, graph: GraphIndex, drop: Dropout, edge_features=None):
        """edge_features: (table [V, F], msg_feat int32 [E]) when the layer was built with features_dimension = F > 0."""
        if self.features_dimension > 0:
            if edge_features is None:
                raise ValueError("this layer was built with features_dimension > 0: the minibatch must carry `msg_feat`")
            table, msg_feat = edge_features
            return hip_ops.mp_layer_with_edge_features(node_states, self.W, self.ln_g, self.ln_b, self.Wd, self.bd, table, msg_feat,
                                                       graph, self._msg_act(), drop)
        return hip_ops.mp_layer(node_states, self.W, self.ln_g, self.ln_b, self.Wd, self.bd, graph, self._msg_act(), drop,
                                aggregation=getattr(self, "message_aggregation_function", "max"))  # (older pickles: max)


class GatedMessagePassingLayer(nn.Module):
    """GGNN layer, kwargs as at the reference call site gnnlayerdefs.py:43-67.  Frozen spec (ptgnn's
    source is unavailable): m_e = h[src] @ W[type(e)] (no bias), max-aggregate
    (0 if none), h' = Dropout(GRUCell(input = aggregate, hidden = h)), torch.nn.GRUCell gate order."""

    def __init__(self, state_dimension: int, message_dimension: int, num_edge_types: int,
                 message_aggregation_function: str = "max", dropout_rate: float = 0.0):
        super().__init__()
        if message_aggregation_function != "max":
            raise NotImplementedError("the HIP path implements the reference's `max` aggregation (gnnlayerdefs.py:47,63)")
        D, Dm, T = state_dimension, message_dimension, num_edge_types
        _check_message_width(Dm, "GatedMessagePassingLayer")
        self.state_dimension, self.message_dimension, self.output_state_dimension = D, Dm, D
        self.num_edge_types, self.dropout_rate = T, dropout_rate
        k = 1.0 / math.sqrt(D)
        self.W = nn.Parameter(_uniform_(torch.empty(T, D, Dm), 1.0 / math.sqrt(D)))
        self.Wi = nn.Parameter(_uniform_(torch.empty(Dm, 3 * D), k))  # nn.GRUCell init: U(-1/sqrt(hidden), +)
        self.bi = nn.Parameter(_uniform_(torch.empty(3 * D), k))
        self.Wh = nn.Parameter(_uniform_(torch.empty(D, 3 * D), k))
        self.bh = nn.Parameter(_uniform_(torch.empty(3 * D), k))

    def forward(self, node_states, graph: GraphIndex, drop: Dropout):
        return hip_ops.gated_mp_layer(node_states, self.W, self.Wi, self.bi, self.Wh, self.bh, graph, drop)


class ConcatResidualLayer:
    """Stateless marker pair (gnnlayerdefs.py:24-38): `pass_through_dummy_layer()` stashes the
    current node states, the layer itself returns [stash ; current]."""

    def __init__(self, hidden_state_size: int):
        self.hidden_state_size = hidden_state_size
        self.output_state_dimension = 2 * hidden_state_size
        self.dummy = _PassThrough(self)

    def pass_through_dummy_layer(self):
        return self.dummy


class _PassThrough:
    def __init__(self, owner: ConcatResidualLayer):
        self.owner = owner
        self.output_state_dimension = owner.hidden_state_size


class GraphNeuralNetwork(nn.Module):
    """Embeds nodes then applies the layer recipe.  Called as `gnn(**graph_data, return_all_states=bool)`
    (reference gnn.py:117)."""

    def __init__(self, node_embedder: SubtokenEmbedder, layer_recipe: List[Any], edge_embedder: Optional[TokenEmbedder] = None):
        super().__init__()
        self.embed = node_embedder
        self.edge_embed = edge_embedder  # edge features (modelregistry.py:70-86) or None
        self._recipe = layer_recipe
        mp, seen = [], set()
        for l in layer_recipe:  # a layer object may appear several times (weight sharing, ggnn recipe)
            if isinstance(l, (MlpMessagePassingLayer, GatedMessagePassingLayer)) and id(l) not in seen:
                seen.add(id(l))
                mp.append(l)
        self.mp = nn.ModuleList(mp)
        self.input_node_state_dim = node_embedder.embedding_size
        self.output_node_state_dim = [l for l in layer_recipe if isinstance(l, nn.Module)][-1].output_state_dimension

    @property
    def message_passing_layers(self):
        return list(self.mp)

    def forward(self, *, token_ids, token_lens, msg_src, msg_tgt, type_ptr, tgt_ptr, tgt_msgs, src_ptr, src_msgs,
                node_to_graph, reference_node_ids, reference_node_graph_idx, num_graphs, num_nodes, num_messages,
                return_all_states: bool = False, dropout_seed: Optional[int] = None, tok_occ=None, tok_chunk_ptr=None,
                tok_chunk_id=None, node_order=None, num_hub_nodes: int = -1, msg_feat=None, **_unused) -> GnnOutput:
        graph = GraphIndex(msg_src, msg_tgt, type_ptr, tgt_ptr, tgt_msgs, src_ptr, src_msgs, int(num_nodes),
                           int(num_messages), int(type_ptr.shape[0]) - 1, node_order, int(num_hub_nodes))
        training = self.training and dropout_seed is not None
        seed = int(dropout_seed or 0)
        mk = lambda rate, stream: Dropout(rate if training else 0.0, seed, stream)
        tok_csr = (tok_occ, tok_chunk_ptr, tok_chunk_id) if tok_occ is not None else None
        h0 = self.embed(token_ids, token_lens, mk(self.embed.dropout_rate, 0), tok_csr)
        edge_features = None
        if self.edge_embed is not None:
            if msg_feat is None:
                raise ValueError("the model has an edge-feature embedding but the minibatch carries no `msg_feat`")
            edge_features = (self.edge_embed.table, msg_feat)
        h = h0
        all_states = [h0]
        stash: Dict[int, torch.Tensor] = {}
        li = 0
        for layer in self._recipe:
            if isinstance(layer, _PassThrough):
                stash[id(layer.owner)] = h
            elif isinstance(layer, ConcatResidualLayer):
                # [stash ; current] stays a PAIR: the MLP message-passing layer reads both halves in place
                # (no concatenated copy, and backward writes the two gradients separately)
                h = (stash.pop(id(layer)), h)
            else:
                if isinstance(h, tuple) and not isinstance(layer, MlpMessagePassingLayer):
                    h = torch.cat(h, dim=-1)
                if isinstance(layer, MlpMessagePassingLayer) and layer.features_dimension > 0:
                    h = layer(h, graph, mk(layer.dropout_rate, 1 + li), edge_features)
                else:
                    h = layer(h, graph, mk(layer.dropout_rate, 1 + li))
                li += 1
                all_states.append(h)
        if isinstance(h, tuple):
            h = torch.cat(h, dim=-1)
        out = torch.cat(all_states, dim=-1) if return_all_states else h
        return GnnOutput(h0, out, node_to_graph, reference_node_ids, reference_node_graph_idx, int(num_graphs))

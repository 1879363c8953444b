"""Host-side graph tensoriser: the roles ptgnn's `StrElementRepresentationModel` and
`GraphNeuralNetworkModel` play in the reference (kwargs pinned at buglab/models/modelregistry.py:71-90;
methods used at buglab/models/gnn.py:350,354,403,433,466,547).  Metadata pass -> vocabularies and
edge-type table; `tensorize` -> int32 NumPy arrays per graph (buglab.data.collate.TensorizedGraphData).

Choices ptgnn's call sites do not pin, frozen here: subtoken vocabulary
built with count threshold 5; presented edge types = sorted forward types, then (if
`add_backwards_edges`, default True) one reversed type per forward type, then (if
`add_self_edges`) one self-loop type -- e.g. BugLab's 15 forward kinds -> 31 presented types.
"""
from __future__ import annotations

from collections import Counter
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from buglab.data.collate import TensorizedGraphData
from buglab.representations.data import GraphData
from buglab.runtime.vocabulary import Vocabulary, split_identifier_into_parts


class StrElementRepresentationModel:
    def __init__(self, *, token_splitting: str = "subtoken", embedding_size: int = 128, vocabulary_size: int = 15000,
                 max_num_subtokens: int = 6, subtoken_combination: str = "max", dropout_rate: float = 0.0,
                 min_freq_threshold: int = 5, dropout_placement: str = "after_pooling"):
        if token_splitting not in ("subtoken", "token"):
            raise NotImplementedError("the HIP embedders implement subtoken (the node model, modelregistry.py:61-67) and "
                                      "token (the edge-feature model, modelregistry.py:70-74) splitting")
        if subtoken_combination not in ("max", "sum", "mean"):
            raise ValueError(f"subtoken_combination must be 'max', 'sum' or 'mean' (got {subtoken_combination!r})")
        self.token_splitting, self.subtoken_combination = token_splitting, subtoken_combination
        self.embedding_size, self.vocabulary_size = embedding_size, vocabulary_size
        self.max_num_subtokens, self.dropout_rate = max_num_subtokens, dropout_rate
        self.min_freq_threshold = min_freq_threshold
        # "after_pooling" (default) / "before_pooling": where the subtoken embedder's dropout sits relative to the max over
        # subtokens -- the second point ptgnn leaves unpinned; reaches gnn() through `node_representations`
        self.dropout_placement = dropout_placement
        self._counter: Optional[Counter] = Counter()
        self.vocabulary: Optional[Vocabulary] = None
        self._cache: Dict[str, np.ndarray] = {}
        self._native_vocab = None  # buglab.data.native.NativeVocabulary, built on first use

    def update_metadata_from(self, node_str: str) -> None:
        if self.token_splitting == "token":
            self._counter[node_str] += 1  # whole strings: the edges' feature tokens (modelregistry.py:70-74)
        else:
            self._counter.update(split_identifier_into_parts(node_str))

    def finalize_metadata(self) -> None:
        self.vocabulary = Vocabulary.create_vocabulary(self._counter, max_size=self.vocabulary_size,
                                                       count_threshold=self.min_freq_threshold, add_unk=True, add_pad=True)
        self._counter = None

    def build_neural_module(self):
        from buglab.models.layers.messagepassing import SubtokenEmbedder, TokenEmbedder

        if self.token_splitting == "token":
            return TokenEmbedder(len(self.vocabulary), self.embedding_size)
        return SubtokenEmbedder(len(self.vocabulary), self.embedding_size, self.max_num_subtokens, self.dropout_rate,
                                getattr(self, "dropout_placement", "after_pooling"), getattr(self, "subtoken_combination", "max"))

    def tensorize_tokens(self, strs) -> np.ndarray:
        """token mode: one vocabulary id per string (the pad token is a vocabulary entry; unseen strings -> unk)."""
        assert self.token_splitting == "token"
        get = self.vocabulary.get_id_or_unk
        return np.fromiter((get(s) for s in strs), dtype=np.int32, count=len(strs))

    def tensorize_nodes(self, node_strs: List[str]):
        S = self.max_num_subtokens
        from buglab.data.native import NativeNodes, NativeVocabulary

        if isinstance(node_strs, NativeNodes) and self.token_splitting == "subtoken":
            # node strings still in the native reader's blob: split + look up in C++ (bl_tensorize_nodes)
            if self._native_vocab is None:
                self._native_vocab = NativeVocabulary(self.vocabulary.id_to_token, self.vocabulary.get_id_or_unk(self.vocabulary.get_unk()))
            ids, lens, needs_python = self._native_vocab.tensorize(node_strs, S)
            for i in np.flatnonzero(needs_python).tolist():  # non-ASCII strings: Unicode lower-casing in Python
                row = [self.vocabulary.get_id_or_unk(t) for t in split_identifier_into_parts(node_strs[i])[:S]]
                ids[i, : len(row)] = row
                lens[i] = max(1, len(row))
            return ids, lens
        ids = np.zeros((len(node_strs), S), dtype=np.int32)
        lens = np.ones(len(node_strs), dtype=np.int32)
        for i, s in enumerate(node_strs):
            row = self._cache.get(s)
            if row is None:
                toks = split_identifier_into_parts(s)[:S]
                row = np.array([self.vocabulary.get_id_or_unk(t) for t in toks], dtype=np.int32)
                if len(self._cache) < 200000:
                    self._cache[s] = row
            ids[i, : len(row)] = row
            lens[i] = max(1, len(row))
        return ids, lens

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_cache"] = {}
        d["_native_vocab"] = None
        return d


class GraphNeuralNetworkModel:
    def __init__(self, *, node_representation_model: StrElementRepresentationModel, edge_representation_model=None,
                 add_self_edges: bool = False, add_backwards_edges: bool = True,
                 message_passing_layer_creator: Callable[[int], List[Any]] = None,
                 stop_extending_minibatch_after_num_nodes: int = 30000, max_nodes_per_graph: int = 35000):
        # edge features (modelregistry.py:70-86): a token-level vocabulary over the edges' third elements; every message carries
        # the id of its edge's token -- reversed edges their forward edge's, self loops the pad token's (frozen here: ptgnn's
        # source is unavailable)
        self.edge_representation_model = edge_representation_model
        self.node_representation_model = node_representation_model
        self.add_self_edges, self.add_backwards_edges = add_self_edges, add_backwards_edges
        self.message_passing_layer_creator = message_passing_layer_creator
        self.stop_extending_minibatch_after_num_nodes = stop_extending_minibatch_after_num_nodes
        self.max_nodes_per_graph = max_nodes_per_graph
        self._edge_types_seen = set()
        self.edge_types: Optional[List[str]] = None

    # metadata
    def update_metadata_from(self, graph: GraphData) -> None:
        for node in graph.node_information:
            self.node_representation_model.update_metadata_from(node)
        self._edge_types_seen.update(graph.edges.keys())
        if self.edge_representation_model is not None and graph.edge_features is not None:
            for feats in graph.edge_features.values():
                for f in feats:
                    self.edge_representation_model.update_metadata_from(f)

    def finalize_metadata(self) -> None:
        self.node_representation_model.finalize_metadata()
        if self.edge_representation_model is not None:
            self.edge_representation_model.finalize_metadata()
        self.edge_types = sorted(self._edge_types_seen)
        self._edge_types_seen = None

    @property
    def num_presented_edge_types(self) -> int:
        n = len(self.edge_types)
        return n * (2 if self.add_backwards_edges else 1) + (1 if self.add_self_edges else 0)

    def build_neural_module(self):
        from buglab.models.layers.messagepassing import GraphNeuralNetwork

        edge_embedder = self.edge_representation_model.build_neural_module() if self.edge_representation_model is not None else None
        return GraphNeuralNetwork(self.node_representation_model.build_neural_module(),
                                  self.message_passing_layer_creator(self.num_presented_edge_types), edge_embedder=edge_embedder)

    def tensorize(self, graph: GraphData) -> Optional[TensorizedGraphData]:
        n = len(graph.node_information)
        if n > self.max_nodes_per_graph:
            return None
        ids, lens = self.node_representation_model.tensorize_nodes(graph.node_information)
        empty = np.zeros((0, 2), dtype=np.int32)
        fwd = [np.asarray(graph.edges.get(t, empty), dtype=np.int32).reshape(-1, 2) for t in self.edge_types]
        adj = list(fwd)
        if self.add_backwards_edges:
            adj += [np.ascontiguousarray(a[:, ::-1]) for a in fwd]
        if self.add_self_edges:
            ar = np.arange(n, dtype=np.int32)
            adj.append(np.stack([ar, ar], axis=1))
        refs = {k: np.asarray(v, dtype=np.int32) for k, v in graph.reference_nodes.items()}
        feat_ids = None
        if self.edge_representation_model is not None:
            em = self.edge_representation_model
            pad = Vocabulary.get_pad()
            feats = graph.edge_features or {}
            f_fwd = []
            for t, a in zip(self.edge_types, fwd):
                f = feats.get(t)
                f_fwd.append(em.tensorize_tokens(list(f)) if f is not None and len(f) == a.shape[0] else em.tensorize_tokens([pad] * a.shape[0]))
            feat_ids = list(f_fwd)
            if self.add_backwards_edges:
                feat_ids += [f.copy() for f in f_fwd]
            if self.add_self_edges:
                feat_ids.append(em.tensorize_tokens([pad] * n))
        return TensorizedGraphData(ids, lens, adj, refs, feat_ids)

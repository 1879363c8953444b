"""Graph data types the collator consumes -- counterpart of reference
buglab/representations/data.py:14-20 (BugLabGraph), :97-121 (HasSubtoken open-vocabulary edges),
:130-167 (BugLabData.as_graph_data).  `.dot` export and the type-annotation task are out of scope."""
import re
from typing import Any, Dict, List, NamedTuple, Optional, Tuple, Union

from collections.abc import Mapping

import numpy as np
from typing_extensions import TypedDict

from buglab.runtime.vocabulary import Vocabulary, split_identifier_into_parts


class GraphData(NamedTuple):
    """Field names pinned by the reference call site data.py:152-165 (ptgnn.GraphData)."""

    node_information: List[str]
    edges: Dict[str, np.ndarray]  # edge type -> int32 [E, 2]
    reference_nodes: Dict[str, Any]
    edge_features: Optional[Dict[str, List]] = None


class BugLabGraph(TypedDict):
    nodes: List[str]
    edges: Dict[str, List[Union[Tuple[int, int], Tuple[int, int, str]]]]
    path: str
    text: str
    reference_nodes: List[int]
    code_range: Tuple[Tuple[int, int], Tuple[int, int]]


IS_IDENTIFIER = re.compile(r"[a-zA-Z_][a-zA-Z0-9_]*")


def add_open_vocab_nodes_and_edges(graph: BugLabGraph) -> None:
    """reference data.py:97-121: one node per distinct subtoken of identifier tokens, linked by
    `HasSubtoken` edges (mutates the graph in place, as the reference does).  Like the reference, the token
    nodes are visited in the iteration order of a Python `set` (data.py:109): that order numbers the subtoken
    nodes and orders the HasSubtoken edges, so it is part of the data contract (the native reader restates
    CPython's set layout, csrc_data/bl_data.cpp::cpython_int_set_order)."""
    if "NextToken" not in graph["edges"]:
        return
    token_nodes = set()
    for edge in graph["edges"]["NextToken"]:
        token_nodes.add(int(edge[0]))  # (plain ints also when the edge list is an int32 array: same hashes, same order)
        token_nodes.add(int(edge[1]))
    vocab_nodes: Dict[str, int] = {}
    vocab_edges: List[Tuple[int, int]] = []
    all_nodes = graph["nodes"]
    for node_idx in token_nodes:
        token_str = all_nodes[node_idx]
        if not IS_IDENTIFIER.match(token_str):
            continue
        for subtoken in split_identifier_into_parts(token_str):
            subtoken_node_idx = vocab_nodes.get(subtoken)
            if subtoken_node_idx is None:
                subtoken_node_idx = len(all_nodes)
                all_nodes.append(subtoken)
                vocab_nodes[subtoken] = subtoken_node_idx
            vocab_edges.append((node_idx, subtoken_node_idx))
    graph["edges"]["HasSubtoken"] = vocab_edges


def _as_np_array(arr):
    if len(arr) == 0:
        return np.zeros((0, 2), dtype=np.int32)
    return np.array(arr, dtype=np.int32)


class _NativeEdgeFeatures(Mapping):
    """`GraphData.edge_features` of a natively read graph: edge kind -> the edges' third elements (pad where an edge has
    none), built on demand -- only a model with `edge_feature_size > 0` ever looks (reference data.py:158-161)."""

    def __init__(self, edges):
        self._edges = edges

    def __getitem__(self, kind):
        arr = self._edges.arrays[kind]
        f = self._edges._feats.get(kind)
        pad = Vocabulary.get_pad()
        if f is None:
            return [pad] * int(arr.shape[0])
        strings = self._edges._feat_strings
        return [strings[int(i)] if i >= 0 else pad for i in f.tolist()]

    def __iter__(self):
        return iter(self._edges.arrays)

    def __len__(self):
        return len(self._edges.arrays)


class BugLabData(TypedDict):
    graph: BugLabGraph
    candidate_rewrites: List[Tuple[str, Any]]
    candidate_rewrite_metadata: List[Tuple[str, Any]]
    candidate_rewrite_ranges: List[Tuple[Tuple[int, int], Tuple[int, int]]]
    target_fix_action_idx: Optional[int]
    package_name: str
    candidate_rewrite_logprobs: Optional[List[float]]

    @classmethod
    def as_graph_data(cls, data: "BugLabData") -> Tuple[GraphData, Optional[int]]:
        """reference data.py:139-167."""
        from buglab.data.native import NativeGraph

        if isinstance(data["graph"], NativeGraph):
            # read by the native reader (buglab/data/native.py): subtoken nodes / HasSubtoken edges are already
            # there and the edge lists already are int32 [E, 2] arrays
            g = data["graph"]
            candidate_node_idxs, inv = np.unique(g.reference_nodes_array, return_inverse=True)
            target_node_idx = None
            if data["target_fix_action_idx"] is not None:
                target_node_idx = int(inv[data["target_fix_action_idx"]])
            return (GraphData(node_information=g.nodes, edges=dict(g.edges.arrays), edge_features=_NativeEdgeFeatures(g.edges),
                              reference_nodes={"candidate_nodes": candidate_node_idxs.astype(np.int32)}), target_node_idx)
        candidate_node_idxs, inv = np.unique(data["graph"]["reference_nodes"], return_inverse=True)
        if data["target_fix_action_idx"] is not None:
            target_node_idx = int(inv[data["target_fix_action_idx"]])
            assert data["graph"]["reference_nodes"][data["target_fix_action_idx"]] == candidate_node_idxs[target_node_idx]
        else:
            target_node_idx = None
        if "HasSubtoken" not in data["graph"]["edges"]:
            add_open_vocab_nodes_and_edges(data["graph"])
        return (
            GraphData(
                node_information=data["graph"]["nodes"],
                edges={e_type: _as_np_array([(e[0], e[1]) for e in adj]) for e_type, adj in data["graph"]["edges"].items()},
                edge_features={e_type: [e[2] if len(e) >= 3 else Vocabulary.get_pad() for e in adj]
                               for e_type, adj in data["graph"]["edges"].items()},
                reference_nodes={"candidate_nodes": candidate_node_idxs.astype(np.int32)},
            ),
            target_node_idx,
        )

"""Minibatch collator for the MI355X message-passing path.

Replaces the per-element Python list appends of the reference's
`GnnBugLabModel.initialize/extend/finalize_minibatch`
(reference buglab/models/gnn.py:431-604) and of ptgnn's
`GraphNeuralNetworkModel.*_minibatch` with NumPy array concatenation, and emits
what the HIP kernels want instead of ptgnn's per-type `(int64[E_t], int64[E_t])`
adjacency lists:

  msg_src, msg_tgt : int32[E]   messages grouped TYPE-MAJOR, sorted by target inside a type
  type_ptr         : int32[T+1] extent of each edge type in that order (the GEMM groups)
  tgt_ptr/tgt_msgs : int32[N+1]/int32[E]  CSR node -> ids of its incoming messages (ascending)
  src_ptr/src_msgs : int32[N+1]/int32[E]  CSR node -> ids of the messages it is the source of

Target-sorted CSR turns the reference's atomic `scatter_max` (torch_scatter) into
a segmented reduction: no atomics, deterministic, ties resolve to the lowest
message id.  Everything is int32 (the reference ships int64) and all index
arrays of a minibatch travel to the device in ONE pinned blob / ONE H2D copy
(`to_device`), instead of ~15 `torch.tensor(list)` copies (gnn.py:549-604).

The keys of the finished minibatch dict are the ones `GnnBugLabModule.forward`
takes in the reference (gnn.py:144-167), so `nn(**minibatch)` is unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, NamedTuple, Optional, Sequence

import os

import numpy as np

I32 = np.int32


@dataclass
class TensorizedGraphData:
    """Per-graph arrays (the role of ptgnn's TensorizedGraphData; reference gnn.py:30,478)."""

    token_ids: np.ndarray  # int32 [n, S]   subtoken ids, padded
    token_lens: np.ndarray  # int32 [n]
    adjacency_lists: List[np.ndarray]  # per presented edge type: int32 [E_t, 2] (src, tgt)
    reference_nodes: Dict[str, np.ndarray] = field(default_factory=dict)
    edge_feature_ids: Optional[List[np.ndarray]] = None  # per presented edge type: int32 [E_t] feature-token ids (edge features on)

    @property
    def num_nodes(self) -> int:
        return int(self.token_ids.shape[0])

    @property
    def num_messages(self) -> int:
        return int(sum(a.shape[0] for a in self.adjacency_lists))


class BaseTensorizedBugLabGnn(NamedTuple):
    """Same fields as the reference's NamedTuple (gnn.py:29-52)."""

    graph_data: TensorizedGraphData
    target_location_node_idx: Optional[int]
    target_rewrites: Sequence[int]
    target_rewrite_to_location_group: Sequence[int]
    correct_rewrite_target: Optional[int]
    text_rewrite_original_idx: Sequence[int]
    candidate_symbol_to_varmisused_node: Sequence[int]
    correct_candidate_symbol_node: Optional[int]
    candidate_rewrite_original_idx: Sequence[int]
    swapped_pair_to_call: Sequence[int]
    correct_swapped_pair: Optional[int]
    pair_rewrite_original_idx: Sequence[int]
    num_rewrite_locations_considered: int
    rewrite_logprobs: Optional[Sequence[float]]


REFERENCE_KEYS_1D = (
    "candidate_nodes",
    "target_rewrite_nodes",
    "varmisused_node_ids",
    "candidate_symbol_node_ids",
    "call_node_ids",
)
REFERENCE_KEY_PAIRS = "candidate_swapped_node_ids"


def _csr(keys: np.ndarray, n: int):
    """CSR over `keys` (values in [0, n)): ptr int32[n+1], items int32[len(keys)] ascending inside a segment."""
    from buglab.data.native import counting_sort  # one native pass when libbuglab_data is built, NumPy otherwise

    return counting_sort(keys, n)


def segments_from_index(index: np.ndarray, num_segments: int):
    """CSR (ptr, items) for an arbitrary (unsorted) segment-id vector; used for the repair
    log-softmax whose ids are the concatenation of three per-scout lists (gnn.py:296-298)."""
    return _csr(np.asarray(index, dtype=np.int64), num_segments)


# reference-node families the scoring heads gather (localizationmodule.py:54-60, fixermodules.py:31-39, 65-73, 110-124)
HEAD_REFERENCE_KEYS = ("candidate_nodes", "target_rewrite_nodes", "varmisused_node_ids", "candidate_symbol_node_ids",
                       "call_node_ids", "candidate_swapped_a", "candidate_swapped_b")
HUB_DEGREE = 32  # nodes with more incident messages than this are processed first by the per-node kernels
TOKEN_CHUNK = 256  # occurrences of one token summed by one wave of the embedding-gradient kernel


def _token_chunk() -> int:
    """BL_DETERMINISTIC=1: one chunk per token, so that every embedding row receives exactly one add (no order to fix)."""
    import os

    return (1 << 30) if os.environ.get("BL_DETERMINISTIC", "0") not in ("", "0") else TOKEN_CHUNK



def token_occurrence_chunks(token_ids: np.ndarray, token_lens: np.ndarray, chunk: int = TOKEN_CHUNK):
    """Token-sorted list of the valid subtoken slots, cut in chunks of one token each.

    -> occ int32 [n_occ] (= node * S + slot), chunk_ptr int32 [C + 1], chunk_tok int32 [C].  The embedding
    gradient (`bl_embed_subtoken_max_bwd_sorted`) sums a chunk in registers and issues one atomic per
    channel: subtoken frequencies are Zipfian, and same-address atomics serialise."""
    N, S = token_ids.shape
    valid = np.arange(S)[None, :] < token_lens[:, None]
    flat = np.flatnonzero(valid.reshape(-1))
    ids = token_ids.reshape(-1)[flat]
    if ids.size == 0:
        return np.zeros(0, dtype=I32), np.zeros(1, dtype=I32), np.zeros(0, dtype=I32)
    from buglab.data.native import counting_sort

    tptr, order = counting_sort(ids, int(ids.max()) + 1)
    occ = flat[order].astype(I32)
    counts_all = np.diff(tptr)
    uniq = np.flatnonzero(counts_all)
    start, counts = tptr[:-1][uniq].astype(np.int64), counts_all[uniq].astype(np.int64)
    nch = (counts + chunk - 1) // chunk
    first = np.arange(int(nch.sum())) - np.repeat(np.cumsum(nch) - nch, nch)
    chunk_start = np.repeat(start, nch) + first * chunk
    chunk_ptr = np.concatenate([chunk_start, [occ.size]]).astype(I32)
    return occ, chunk_ptr, np.repeat(uniq, nch).astype(I32)


def _collate_graph_arrays_numpy(graphs, num_edge_types: int, node_off: np.ndarray, N: int) -> Dict[str, np.ndarray]:
    """NumPy version of `buglab.data.native.collate_graph_arrays` (used when the native library is not built, and as
    its test reference)."""
    S = max((g.token_ids.shape[1] for g in graphs), default=1)
    token_ids = np.zeros((N, S), dtype=I32)
    token_lens = np.zeros(N, dtype=I32)
    for g, o in zip(graphs, node_off[:-1]):
        token_ids[o : o + g.num_nodes, : g.token_ids.shape[1]] = g.token_ids
        token_lens[o : o + g.num_nodes] = g.token_lens

    tok_occ, tok_chunk_ptr, tok_chunk_id = token_occurrence_chunks(token_ids, token_lens, _token_chunk())

    # all messages of the batch at once: per graph one concatenation (edge lists in type order), then two stable
    # counting sorts -- by target, then by type -- give "type-major, target-sorted inside a type" with ties in
    # (graph, list) order, exactly what sorting each type's concatenated list by target gives
    srcs, tgts, typs, feats = [], [], [], []
    with_feat = any(g.edge_feature_ids is not None for g in graphs)  # edge features: a feature-token id travels with every message
    for g, o in zip(graphs, node_off[:-1]):
        lists = [a for a in g.adjacency_lists[:num_edge_types]]
        counts = [a.shape[0] for a in lists]
        if sum(counts) == 0:
            continue
        adj = np.concatenate(lists, axis=0).astype(I32, copy=False)
        srcs.append(adj[:, 0] + I32(o))
        tgts.append(adj[:, 1] + I32(o))
        typs.append(np.repeat(np.arange(len(lists), dtype=I32), counts))
        if with_feat:
            assert g.edge_feature_ids is not None and [f.shape[0] for f in g.edge_feature_ids[:num_edge_types]] == counts, \
                "every graph of the minibatch needs one feature id per edge"
            feats.append(np.concatenate(g.edge_feature_ids[:num_edge_types]).astype(I32, copy=False))
    msg_feat = None
    if srcs:
        all_src, all_tgt, all_typ = np.concatenate(srcs), np.concatenate(tgts), np.concatenate(typs)
        _, by_tgt = _csr(all_tgt, N)
        type_ptr, by_typ = _csr(all_typ[by_tgt], num_edge_types)
        order = by_tgt[by_typ]
        msg_src, msg_tgt = np.ascontiguousarray(all_src[order], dtype=I32), np.ascontiguousarray(all_tgt[order], dtype=I32)
        if with_feat:
            msg_feat = np.ascontiguousarray(np.concatenate(feats)[order], dtype=I32)
    else:
        msg_src = msg_tgt = np.zeros(0, dtype=I32)
        type_ptr = np.zeros(num_edge_types + 1, dtype=I32)
        if with_feat:
            msg_feat = np.zeros(0, dtype=I32)
    tgt_ptr, tgt_msgs = _csr(msg_tgt, N)
    src_ptr, src_msgs = _csr(msg_src, N)
    # per-node kernels (one wave per node) take hubs first: a node with hundreds of messages keeps its wave busy
    # for about as long as the whole launch lasts, so it has to start at t = 0 (BASELINE config c4)
    deg = np.diff(tgt_ptr).astype(np.int64) + np.diff(src_ptr)
    hubs = np.flatnonzero(deg > HUB_DEGREE)
    if hubs.size:
        hubs = hubs[np.argsort(-deg[hubs], kind="stable")]
        rest = np.ones(N, dtype=bool)
        rest[hubs] = False
        node_order = np.concatenate([hubs, np.flatnonzero(rest)]).astype(I32)
    else:
        node_order = np.arange(N, dtype=I32)

    out = {"token_ids": token_ids, "token_lens": token_lens, "msg_src": msg_src, "msg_tgt": msg_tgt, "type_ptr": np.asarray(type_ptr, dtype=I32),
           "tgt_ptr": tgt_ptr, "tgt_msgs": tgt_msgs, "src_ptr": src_ptr, "src_msgs": src_msgs, "node_order": node_order,
           "tok_occ": tok_occ, "tok_chunk_ptr": tok_chunk_ptr, "tok_chunk_id": tok_chunk_id}
    if msg_feat is not None:
        out["msg_feat"] = msg_feat
    return out


def collate_graphs(graphs: Sequence[TensorizedGraphData], num_edge_types: int) -> Dict[str, Any]:
    """Disjoint union of graphs -> one `graph_data` dict of NumPy arrays."""
    B = len(graphs)
    n_per_graph = np.array([g.num_nodes for g in graphs], dtype=np.int64)
    node_off = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(n_per_graph, out=node_off[1:])
    N = int(node_off[-1])

    from buglab.data import native

    if native.available() and os.environ.get("BUGLAB_NATIVE_COLLATE", "1") != "0":
        arr = native.collate_graph_arrays(graphs, num_edge_types, HUB_DEGREE, _token_chunk())  # one GIL-free native call
    else:
        arr = _collate_graph_arrays_numpy(graphs, num_edge_types, node_off, N)
    token_ids, token_lens, node_order = arr["token_ids"], arr["token_lens"], arr["node_order"]
    tok_occ, tok_chunk_ptr, tok_chunk_id = arr["tok_occ"], arr["tok_chunk_ptr"], arr["tok_chunk_id"]
    msg_src, msg_tgt, type_ptr = arr["msg_src"], arr["msg_tgt"], arr["type_ptr"]
    tgt_ptr, tgt_msgs, src_ptr, src_msgs = arr["tgt_ptr"], arr["tgt_msgs"], arr["src_ptr"], arr["src_msgs"]

    ref_ids: Dict[str, np.ndarray] = {}
    ref_graph: Dict[str, np.ndarray] = {}
    for key in REFERENCE_KEYS_1D:
        ids, gidx = [], []
        for b, (g, o) in enumerate(zip(graphs, node_off[:-1])):
            r = np.asarray(g.reference_nodes.get(key, ()), dtype=np.int64).reshape(-1)
            ids.append(r + o)
            gidx.append(np.full(r.shape[0], b, dtype=np.int64))
        ref_ids[key] = np.concatenate(ids).astype(I32) if ids else np.zeros(0, I32)
        ref_graph[key] = np.concatenate(gidx).astype(I32) if gidx else np.zeros(0, I32)
    pairs, pair_g = [], []
    for b, (g, o) in enumerate(zip(graphs, node_off[:-1])):
        r = np.asarray(g.reference_nodes.get(REFERENCE_KEY_PAIRS, np.zeros((0, 2))), dtype=np.int64).reshape(-1, 2)
        pairs.append(r + o)
        pair_g.append(np.full(r.shape[0], b, dtype=np.int64))
    ref_ids[REFERENCE_KEY_PAIRS] = (np.concatenate(pairs, axis=0) if pairs else np.zeros((0, 2))).astype(I32)
    ref_graph[REFERENCE_KEY_PAIRS] = (np.concatenate(pair_g) if pair_g else np.zeros(0)).astype(I32)
    # the two columns of the swapped-argument pairs as separate row-index vectors: the pair scorer
    # consumes [h[call] ; h[a] ; h[b]] as three gathered sources (fixermodules.py:116-124)
    ref_ids["candidate_swapped_a"] = np.ascontiguousarray(ref_ids[REFERENCE_KEY_PAIRS][:, 0])
    ref_ids["candidate_swapped_b"] = np.ascontiguousarray(ref_ids[REFERENCE_KEY_PAIRS][:, 1])

    # one gather for all scoring heads: every referenced node row, key by key, and where each key's rows
    # land in that compact copy (the heads then index 0..R-1 instead of 0..N-1)
    head_spans, parts, pos = {}, [], 0
    for key in HEAD_REFERENCE_KEYS:
        r = ref_ids[key]
        head_spans[key] = (pos, int(r.shape[0]))
        parts.append(r)
        pos += int(r.shape[0])
    head_gather_idx = np.concatenate(parts).astype(I32) if parts else np.zeros(0, I32)

    node_to_graph = np.repeat(np.arange(B, dtype=I32), n_per_graph)
    cand_graph = ref_graph["candidate_nodes"]
    cand_ptr = np.zeros(B + 1, dtype=np.int64)
    if cand_graph.size:
        np.cumsum(np.bincount(cand_graph, minlength=B), out=cand_ptr[1:])
    # CSR for the localization log-softmax: items = [candidates..., one NO_BUG slot per graph]
    # (localizationmodule.py:66-77: ids = cat(candidate_to_sample_idx, arange(B)))
    loc_ptr, loc_items = _csr(np.concatenate([cand_graph.astype(np.int64), np.arange(B, dtype=np.int64)]), B)
    return {
        **({"msg_feat": arr["msg_feat"]} if "msg_feat" in arr else {}),
        "loc_group_ptr": loc_ptr,
        "loc_group_items": loc_items,
        "token_ids": token_ids,
        "token_lens": token_lens,
        "node_order": node_order,
        "head_gather_idx": head_gather_idx,
        "head_local_idx": np.arange(head_gather_idx.shape[0], dtype=I32),
        "head_spans": head_spans,
        "tok_occ": tok_occ,
        "tok_chunk_ptr": tok_chunk_ptr,
        "tok_chunk_id": tok_chunk_id,
        "msg_src": msg_src,
        "msg_tgt": msg_tgt,
        "type_ptr": type_ptr.astype(I32),
        "tgt_ptr": tgt_ptr,
        "tgt_msgs": tgt_msgs,
        "src_ptr": src_ptr,
        "src_msgs": src_msgs,
        "node_to_graph": node_to_graph,
        "num_nodes_per_graph": n_per_graph.astype(I32),
        "num_graphs": B,
        # how many leading entries of node_order are hubs (more than HUB_DEGREE incident messages): the segmented max gives
        # each of those with a long target segment a whole workgroup (csrc/bl_graph_ops.hip::segment_max_hub_kernel)
        "num_hub_nodes": int(np.count_nonzero(np.diff(tgt_ptr).astype(np.int64) + np.diff(src_ptr) > HUB_DEGREE)),
        "candidate_ptr": cand_ptr.astype(I32),
        "reference_node_ids": ref_ids,
        "reference_node_graph_idx": ref_graph,
    }


def collate_samples(samples: Sequence[BaseTensorizedBugLabGnn], num_edge_types: int) -> Dict[str, Any]:
    """B tensorized samples -> the minibatch dict (NumPy).  Offsets follow reference
    gnn.py:463-542 (`extend_minibatch_with`) exactly; only the mechanism (array ops) differs."""
    gd = collate_graphs([s.graph_data for s in samples], num_edge_types)
    B = len(samples)
    has_bug = np.zeros(B, dtype=np.bool_)
    correct_cand = np.zeros(B, dtype=I32)
    n_cand = np.array([len(s.graph_data.reference_nodes["candidate_nodes"]) for s in samples], dtype=np.int64)
    cand_off = np.concatenate([[0], np.cumsum(n_cand)])
    n_text = np.array([len(s.target_rewrites) for s in samples], dtype=np.int64)
    n_var = np.array([len(s.candidate_symbol_to_varmisused_node) for s in samples], dtype=np.int64)
    n_pair = np.array([len(s.swapped_pair_to_call) for s in samples], dtype=np.int64)
    n_groups = np.array([s.num_rewrite_locations_considered for s in samples], dtype=np.int64)
    text_off = np.concatenate([[0], np.cumsum(n_text)])
    var_off = np.concatenate([[0], np.cumsum(n_var)])
    pair_off = np.concatenate([[0], np.cumsum(n_pair)])
    group_off = np.concatenate([[0], np.cumsum(n_groups)])
    n_rw = n_text + n_var + n_pair
    rw_off = np.concatenate([[0], np.cumsum(n_rw)])

    def cat(parts, dtype=I32):
        parts = [np.asarray(p, dtype=np.int64).reshape(-1) for p in parts]
        return (np.concatenate(parts) if parts else np.zeros(0)).astype(dtype)

    correct_rewrite, correct_symbol, correct_pair = [], [], []
    for b, s in enumerate(samples):
        if s.target_location_node_idx is not None:
            has_bug[b] = True
            correct_cand[b] = s.target_location_node_idx + cand_off[b]  # gnn.py:473-476
        if s.correct_rewrite_target is not None:
            correct_rewrite.append(s.correct_rewrite_target + text_off[b])  # :485-488
        if s.correct_candidate_symbol_node is not None:
            correct_symbol.append(s.correct_candidate_symbol_node + var_off[b])  # :499-503
        if s.correct_swapped_pair is not None:
            correct_pair.append(s.correct_swapped_pair + pair_off[b])  # :513-517
    mb: Dict[str, Any] = {
        "graph_data": gd,
        "has_bug": has_bug,
        "correct_candidate_node_idxs": correct_cand,
        "target_rewrites": cat([s.target_rewrites for s in samples]),
        "rewrite_to_location_group": cat([np.asarray(s.target_rewrite_to_location_group, np.int64) + group_off[b] for b, s in enumerate(samples)]),
        "correct_rewrite_idxs": cat([correct_rewrite]),
        "text_rewrite_idxs": cat([np.asarray(s.text_rewrite_original_idx, np.int64) + rw_off[b] for b, s in enumerate(samples)]),
        "candidate_symbol_to_location_group": cat([np.asarray(s.candidate_symbol_to_varmisused_node, np.int64) + group_off[b] for b, s in enumerate(samples)]),
        "correct_candidate_symbols": cat([correct_symbol]),
        "candidate_rewrite_idxs": cat([np.asarray(s.candidate_rewrite_original_idx, np.int64) + rw_off[b] for b, s in enumerate(samples)]),
        "swapped_pair_to_call_location_group": cat([np.asarray(s.swapped_pair_to_call, np.int64) + group_off[b] for b, s in enumerate(samples)]),
        "correct_swapped_pair": cat([correct_pair]),
        "pair_rewrite_idxs": cat([np.asarray(s.pair_rewrite_original_idx, np.int64) + rw_off[b] for b, s in enumerate(samples)]),
        "rewrite_to_graph_id": np.repeat(np.arange(B, dtype=I32), n_rw),
        # visualisation info kept as python lists (gnn.py:527-530)
        "text_rewrite_original_idxs": [list(s.text_rewrite_original_idx) for s in samples],
        "candidate_rewrite_original_idxs": [list(s.candidate_rewrite_original_idx) for s in samples],
        "pair_rewrite_original_idx": [list(s.pair_rewrite_original_idx) for s in samples],
        "num_repair_groups": int(group_off[-1]),
    }
    # one CSR for the single repair log-softmax over location groups (gnn.py:295-299)
    groups = np.concatenate(
        [mb["rewrite_to_location_group"], mb["candidate_symbol_to_location_group"], mb["swapped_pair_to_call_location_group"]]
    )
    mb["repair_group_ptr"], mb["repair_group_items"] = segments_from_index(groups, mb["num_repair_groups"])
    if any(s.rewrite_logprobs is not None for s in samples):  # gnn.py:536-540, 598-603
        lp = [np.asarray(s.rewrite_logprobs[:-1], dtype=np.float32) for s in samples if s.rewrite_logprobs is not None]
        nb = [np.float32(s.rewrite_logprobs[-1]) for s in samples if s.rewrite_logprobs is not None]
        mb["rewrite_logprobs"] = np.concatenate(lp + [np.asarray(nb, dtype=np.float32)])
        # selector loss (reference utils.py:136-137): CSR, by graph, over the OBSERVED entries (finite
        # detection log-probabilities) of [all rewrites..., one NO_BUG slot per graph]
        index = np.concatenate([mb["rewrite_to_graph_id"].astype(np.int64), np.arange(B, dtype=np.int64)])
        observed = np.flatnonzero(~np.isinf(mb["rewrite_logprobs"]))
        ng = int(index[observed].max()) + 1 if observed.size else 0
        ptr, order = _csr(index[observed], ng)
        mb["gen_group_ptr"], mb["gen_group_items"], mb["gen_num_groups"] = ptr, observed[order].astype(I32), ng
    return mb


_INT_KEYS_MB = (
    "correct_candidate_node_idxs",
    "target_rewrites",
    "rewrite_to_location_group",
    "correct_rewrite_idxs",
    "text_rewrite_idxs",
    "candidate_symbol_to_location_group",
    "correct_candidate_symbols",
    "candidate_rewrite_idxs",
    "swapped_pair_to_call_location_group",
    "correct_swapped_pair",
    "pair_rewrite_idxs",
    "rewrite_to_graph_id",
    "repair_group_ptr",
    "repair_group_items",
)
_INT_KEYS_GD = ("loc_group_ptr", "loc_group_items", "token_ids", "token_lens", "tok_occ", "tok_chunk_ptr", "tok_chunk_id",
                "head_gather_idx", "head_local_idx", "node_order", "msg_src", "msg_tgt", "type_ptr", "tgt_ptr", "tgt_msgs", "src_ptr", "src_msgs", "node_to_graph", "candidate_ptr")


# present only in minibatches of the sequence models (buglab.models.seqmodel.collate_sequences) / with edge features on
_INT_KEYS_GD_SEQ = ("seq_lens", "erow_ptr", "ekey", "ecode", "msg_feat")


def pack_minibatch(mb: Dict[str, Any], out: Optional[np.ndarray] = None):
    """Host half of `to_device`: every int32 array of a collated minibatch laid out in ONE int32 blob (16-byte aligned
    pieces) + the small metadata needed to take it apart again.  -> (blob int32 [total], meta dict).  `out`: write into
    this buffer (e.g. a shared-memory segment of a loader process) instead of allocating; it must hold `packed_size(mb)`."""
    gd = mb["graph_data"]
    arrays = []
    for k in _INT_KEYS_GD + tuple(k for k in _INT_KEYS_GD_SEQ if k in gd):
        arrays.append(("gd", k, np.ascontiguousarray(gd[k], dtype=I32)))
    for k, v in gd["reference_node_ids"].items():
        arrays.append(("ref", k, np.ascontiguousarray(v, dtype=I32)))
    for k, v in gd["reference_node_graph_idx"].items():
        arrays.append(("refg", k, np.ascontiguousarray(v, dtype=I32)))
    for k in _INT_KEYS_MB + (("gen_group_ptr", "gen_group_items") if "gen_group_ptr" in mb else ()):
        arrays.append(("mb", k, np.ascontiguousarray(mb[k], dtype=I32)))
    arrays.append(("mb", "has_bug", np.ascontiguousarray(mb["has_bug"], dtype=I32)))
    layout, total = [], 0
    for where, k, a in arrays:
        layout.append((where, k, tuple(a.shape), total))
        total += (a.size + 3) // 4 * 4  # 16-byte align every array inside the blob
    total = max(total, 4)
    blob = np.empty(total, dtype=I32) if out is None else out[:total]
    for (_, _, a), (_, _, _, o) in zip(arrays, layout):
        blob[o : o + a.size] = a.reshape(-1)
    meta = {
        "layout": layout, "total": total, "head_spans": dict(gd["head_spans"]), "num_graphs": int(gd["num_graphs"]),
        "num_nodes": int(gd["token_ids"].shape[0]), "num_messages": int(gd["msg_src"].shape[0]),
        "type_ptr_host": np.asarray(gd["type_ptr"], dtype=np.int64), "num_nodes_per_graph": np.asarray(gd["num_nodes_per_graph"]),
        "num_repair_groups": int(mb["num_repair_groups"]), "num_hub_nodes": int(gd.get("num_hub_nodes", -1)),
        "original_idxs": {k: mb[k] for k in ("text_rewrite_original_idxs", "candidate_rewrite_original_idxs", "pair_rewrite_original_idx")},
    }
    if "rewrite_logprobs" in mb:
        meta["rewrite_logprobs"] = np.asarray(mb["rewrite_logprobs"], dtype=np.float32)
        meta["gen_num_groups"] = int(mb["gen_num_groups"])
    if "seq_len" in gd:
        meta["seq"] = {"seq_batch": int(gd["seq_batch"]), "seq_len": int(gd["seq_len"])}
        meta["node_mappings"] = mb.get("node_mappings")
    return blob, meta


def packed_size(mb: Dict[str, Any]) -> int:
    """Upper bound (int32 elements) of the blob `pack_minibatch` writes."""
    gd = mb["graph_data"]
    n = sum(int(np.size(gd[k])) + 3 for k in _INT_KEYS_GD + tuple(k for k in _INT_KEYS_GD_SEQ if k in gd))
    n += sum(int(np.size(v)) + 3 for v in gd["reference_node_ids"].values()) + sum(int(np.size(v)) + 3 for v in gd["reference_node_graph_idx"].values())
    n += sum(int(np.size(mb[k])) + 3 for k in _INT_KEYS_MB + (("gen_group_ptr", "gen_group_items") if "gen_group_ptr" in mb else ()))
    return n + int(np.size(mb["has_bug"])) + 8


def upload_packed(blob: np.ndarray, meta: Dict[str, Any], device) -> Dict[str, Any]:
    """Device half of `to_device`: ONE (pinned, non-blocking) host->device copy, then views.

    Host-side copies that kernels' launch geometry needs (`type_ptr_host`, counts) stay as
    Python/NumPy values so no device->host sync is ever needed to size a grid."""
    import torch

    dev = torch.device(device)
    total = int(meta["total"])
    # straight into a pinned buffer from the caching host allocator: a pageable temporary first would be a fresh 16 MB
    # mmap + 4 000 page faults + munmap per minibatch, in a process whose other threads then take the TLB shootdowns
    staging = torch.empty(total, dtype=torch.int32, pin_memory=dev.type == "cuda")
    np.copyto(staging.numpy(), blob[:total])
    dblob = staging.to(dev, non_blocking=True)
    out_gd: Dict[str, Any] = {"reference_node_ids": {}, "reference_node_graph_idx": {}}
    out: Dict[str, Any] = {"graph_data": out_gd}
    for where, k, shape, o in meta["layout"]:
        t = dblob[o : o + int(np.prod(shape, dtype=np.int64))].view(shape)
        if where == "gd":
            out_gd[k] = t
        elif where == "ref":
            out_gd["reference_node_ids"][k] = t
        elif where == "refg":
            out_gd["reference_node_graph_idx"][k] = t
        else:
            out[k] = t
    out["has_bug"] = out["has_bug"].bool()
    out_gd["head_spans"] = dict(meta["head_spans"])
    for k in ("num_graphs", "num_nodes", "num_messages", "type_ptr_host", "num_nodes_per_graph"):
        out_gd[k] = meta[k]
    out_gd["num_hub_nodes"] = meta.get("num_hub_nodes", -1)
    out_gd["_blob"] = dblob  # keeps the single allocation alive
    out["num_repair_groups"] = meta["num_repair_groups"]
    out.update(meta["original_idxs"])
    if "rewrite_logprobs" in meta:
        out["rewrite_logprobs"] = torch.from_numpy(meta["rewrite_logprobs"]).to(dev)
        out["gen_num_groups"] = meta["gen_num_groups"]
    if "seq" in meta:
        out_gd.update(meta["seq"])
        out["node_mappings"] = meta["node_mappings"]
    return out


def to_device(mb: Dict[str, Any], device) -> Dict[str, Any]:
    """NumPy minibatch -> torch tensors on `device` with ONE host->device copy for all int32 arrays."""
    return upload_packed(*pack_minibatch(mb), device)

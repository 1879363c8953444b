"""Rewrite bookkeeping shared by the BugLab models -- counterpart of reference
buglab/models/basemodel.py (`AbstractBugLabModel`): the rewrite-operator vocabulary (:13-69),
grouping of a sample's candidate rewrites by scout and location into flat index arrays (:80-238)
and un-batching of predicted log-probabilities into per-sample results (:240-346).  Host-side
Python; outputs are consumed by buglab.data.collate."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from buglab.runtime.vocabulary import Vocabulary

_ARITH = ["+", "-", "*", "/", "**", "//", "%", "@", "<<", ">>", "|", "&", "^"]
OPERATOR_REWRITES = frozenset(
    _ARITH + [op + "=" for op in _ARITH] + ["=", "<", "<=", ">", ">=", "==", "!=", " in ", " not in ", " is ", " is not ",
                                             "0", "1", "2", "-1", "-2", "and", "or", "not ", "", "True", "False"]
)  # reference basemodel.py:13-64 (48 entries; language specific)


class _FlatSelection:
    """Flattened candidates of one scout family (the reference's `to_flat_node_selection`, :157-182)."""

    __slots__ = ("location_node_ids", "payload", "location_groups", "correct_idx", "original_rewrite_idxs")

    def __init__(self):
        self.location_node_ids: List[int] = []
        self.payload: List = []
        self.location_groups: List[int] = []
        self.correct_idx: Optional[int] = None
        self.original_rewrite_idxs: List[int] = []


class AbstractBugLabModel:
    OPERATOR_REWRITES = OPERATOR_REWRITES

    def _init(self):
        self._target_rewrite_ops = Vocabulary.create_vocabulary(
            sorted(self.OPERATOR_REWRITES), max_size=len(self.OPERATOR_REWRITES), count_threshold=0, add_unk=False)
        self._tensorize_only_at_target_location_rewrites = True

    @contextmanager
    def _tensorize_all_location_rewrites(self):
        try:
            self._tensorize_only_at_target_location_rewrites = False
            yield
        finally:
            self._tensorize_only_at_target_location_rewrites = True

    # -------------------------------------------------------------------------------------------
    def _compute_rewrite_data(self, datapoint, candidate_node_idxs: Sequence[int]):
        """Same 16-tuple as reference basemodel.py:80-238.

        During training only rewrites AT THE TARGET LOCATION are kept (:119-121); with
        `_tensorize_all_location_rewrites()` (predict) every location is kept."""
        graph = datapoint["graph"]
        target_idx = datapoint["target_fix_action_idx"]
        target_node = graph["reference_nodes"][target_idx] if target_idx is not None else None

        call_args: Dict[int, List[int]] = defaultdict(list)  # Call node -> its `args` children, in order (:88-98)
        nodes = graph["nodes"]
        from buglab.data.native import NativeGraph

        if isinstance(graph, NativeGraph):
            # native reader: the labelled Child edges are a few entries of an int32 array -- no Python loop over all edges
            for src, tgt in graph.edges.labelled("Child", "args"):
                if nodes[src] == "Call":
                    call_args[src].append(tgt)
        else:
            for edge in graph["edges"].get("Child", ()):
                if len(edge) == 3 and edge[2] == "args" and nodes[edge[0]] == "Call":
                    call_args[edge[0]].append(edge[1])

        # per family: location node -> (payloads, original rewrite ids); insertion-ordered like the reference's dicts
        fam_payload = {k: defaultdict(list) for k in ("text", "var", "swap")}
        fam_orig = {k: defaultdict(list) for k in ("text", "var", "swap")}
        fam_correct: Dict[str, Optional[Tuple[int, int]]] = {"text": None, "var": None, "swap": None}

        for i, (node_idx, (_rw_type, rw_data), (scout, rw_meta)) in enumerate(
                zip(graph["reference_nodes"], datapoint["candidate_rewrites"], datapoint["candidate_rewrite_metadata"])):
            if self._tensorize_only_at_target_location_rewrites and node_idx != target_node:
                continue
            if scout == "VariableMisuseRewriteScout":
                fam, payload = "var", rw_meta
            elif scout == "ArgSwapRewriteScout":
                args = call_args[node_idx]
                fam, payload = "swap", (args[rw_data[0]], args[rw_data[1]])
            else:
                fam, payload = "text", self._target_rewrite_ops.get_id_or_unk(rw_data)
            if target_idx == i:
                fam_correct[fam] = (node_idx, len(fam_payload[fam][node_idx]))
            fam_payload[fam][node_idx].append(payload)
            fam_orig[fam][node_idx].append(i)

        group_of = {int(n): g for g, n in enumerate(candidate_node_idxs)}  # :155

        def flatten(fam: str) -> _FlatSelection:
            out = _FlatSelection()
            correct = fam_correct[fam]
            for loc_node, payloads in fam_payload[fam].items():
                if correct is not None and correct[0] == loc_node:
                    out.correct_idx = len(out.payload) + correct[1]
                out.location_node_ids.extend([loc_node] * len(payloads))
                out.payload.extend(payloads)
                out.location_groups.extend([group_of[int(loc_node)]] * len(payloads))
                out.original_rewrite_idxs.extend(fam_orig[fam][loc_node])
            return out

        text, var, swap = flatten("text"), flatten("var"), flatten("swap")
        return (
            text.location_node_ids, text.payload, text.location_groups, text.correct_idx, text.original_rewrite_idxs,
            var.location_node_ids, var.location_groups, var.payload, var.correct_idx, var.original_rewrite_idxs,
            swap.location_node_ids, swap.payload, swap.correct_idx, swap.location_groups, swap.original_rewrite_idxs,
            group_of,
        )

    # -------------------------------------------------------------------------------------------
    def _iter_per_sample_results(self, mb_data, candidate_location_sample_idx, candidate_location_log_probs,
                                 arg_swap_logprobs, num_samples, original_datapoints, text_repair_logprobs,
                                 varmisuse_logprobs, node_mappings: List[Dict[int, int]] = None):
        """Un-batch a predicted minibatch into (datapoint, {node_idx: logprob, -1: NO_BUG}, [rewrite logprob])
        triples -- reference basemodel.py:240-346.  All array work is NumPy; one D2H per tensor."""
        to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        loc_sample = to_np(candidate_location_sample_idx)
        loc_lp = to_np(candidate_location_log_probs)
        per_sample_loc = [loc_lp[loc_sample == b] for b in range(num_samples)]  # candidates in order, NO_BUG last

        def by_group(logprobs, groups):
            d = defaultdict(list)
            for g, lp in zip(to_np(groups).tolist(), to_np(logprobs).tolist()):
                d[g].append(lp)
            return d

        swap_g = by_group(arg_swap_logprobs, mb_data["swapped_pair_to_call_location_group"])
        text_g = by_group(text_repair_logprobs, mb_data["rewrite_to_location_group"])
        var_g = by_group(varmisuse_logprobs, mb_data["candidate_symbol_to_location_group"])

        next_group = 0
        for b in range(num_samples):
            point = original_datapoints[b]
            ref_nodes = point["graph"]["reference_nodes"]
            cand_nodes = np.unique(ref_nodes)
            if node_mappings is not None:
                cand_nodes = np.array([node_mappings[b][k] for k in cand_nodes])
            dist = per_sample_loc[b]
            assert len(dist) == len(cand_nodes) + 1
            location_logprobs = {int(n): float(lp) for n, lp in zip(cand_nodes, dist)}
            location_logprobs[-1] = float(dist[-1])

            flat_swap, flat_text, flat_var = [], [], []
            for _ in range(len(np.unique(ref_nodes))):
                flat_swap.extend(swap_g[next_group])
                flat_text.extend(text_g[next_group])
                flat_var.extend(var_g[next_group])
                next_group += 1
            text_idx = mb_data["text_rewrite_original_idxs"][b]
            var_idx = mb_data["candidate_rewrite_original_idxs"][b]
            swap_idx = mb_data["pair_rewrite_original_idx"][b]
            assert len(text_idx) == len(flat_text) and len(var_idx) == len(flat_var) and len(swap_idx) == len(flat_swap)
            rewrite_probs: List[Optional[float]] = [None] * len(point["candidate_rewrites"])
            for idxs, lps in ((text_idx, flat_text), (var_idx, flat_var), (swap_idx, flat_swap)):
                for i, lp in zip(idxs, lps):
                    assert rewrite_probs[i] is None
                    rewrite_probs[i] = lp
            assert None not in rewrite_probs

            if node_mappings is not None:  # :320-335 (sequence models map graph nodes to tokens)
                reverse = defaultdict(list)
                for old, new in node_mappings[b].items():
                    if old in ref_nodes:
                        reverse[new].append(old)
                remapped = {}
                for n, p in location_logprobs.items():
                    for node in (reverse[n] if n >= 0 else [n]):
                        remapped[node] = p
                location_logprobs = remapped
            yield point, location_logprobs, rewrite_probs

"""Seeded synthetic "PyPI-shaped" code graphs.

There is no network for the real PyPIBugs/RandomBugs shards, so bench and
parity runs use graphs with the same *shape*: N nodes, E directed messages split
over T presented edge types with Zipf(1.0) shares (a few dominant kinds such as
Child/NextToken, a long tail), uniform or truncated power-law in-degree,
Zipf subtoken ids over the 15000-entry vocabulary with 1-6 subtokens per node,
40 candidate bug locations per graph, half of the graphs buggy, and for a buggy
graph 4 text-rewrite, 6 variable-misuse and 3 argument-swap candidates at the
target location (one of the 13 correct) -- the field layout is the reference's
`BaseTensorizedBugLabGnn` (buglab/models/gnn.py:29-52).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from buglab.data.collate import BaseTensorizedBugLabGnn, TensorizedGraphData

I32 = np.int32


def _zipf_probs(n: int, a: float = 1.0) -> np.ndarray:
    w = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** a
    return w / w.sum()


def _powerlaw_targets(rng, n_nodes: int, n_msgs: int, alpha: float, max_degree: int) -> np.ndarray:
    """Targets whose in-degree follows a truncated power law (config c4: alpha 2.0, max 512)."""
    deg = np.arange(1, max_degree + 1, dtype=np.float64)
    pk = deg**-alpha
    pk /= pk.sum()
    d = rng.choice(np.arange(1, max_degree + 1), size=n_nodes, p=pk).astype(np.int64)
    d[0] = max_degree  # make sure the hub case is always present
    # rescale to n_msgs while keeping the shape
    tgt = np.repeat(np.arange(n_nodes), d)
    if tgt.size >= n_msgs:
        tgt = rng.permutation(tgt)[:n_msgs]
        hub = np.zeros(min(max_degree, n_msgs), dtype=np.int64)
        tgt[: hub.size] = hub
    else:
        tgt = np.concatenate([tgt, rng.integers(0, n_nodes, size=n_msgs - tgt.size)])
    return tgt


def make_sample(
    rng: np.random.Generator,
    num_nodes: int = 2000,
    num_messages: int = 10000,
    num_edge_types: int = 16,
    vocab_size: int = 15000,
    max_subtokens: int = 6,
    num_candidates: int = 40,
    rewrite_vocab_size: int = 48,
    buggy: Optional[bool] = None,
    degree: str = "uniform",
    max_degree: int = 512,
    n_text: int = 4,
    n_var: int = 6,
    n_swap: int = 3,
) -> BaseTensorizedBugLabGnn:
    n, E, T = num_nodes, num_messages, num_edge_types
    lens = np.minimum(1 + rng.geometric(0.5, size=n) - 1, max_subtokens).astype(I32)  # mean ~2
    lens = np.maximum(lens, 1)
    ids = rng.choice(vocab_size, size=(n, max_subtokens), p=_zipf_probs(vocab_size)).astype(I32)
    ids[np.arange(max_subtokens)[None, :] >= lens[:, None]] = 0

    etype = rng.choice(T, size=E, p=_zipf_probs(T))
    src = rng.integers(0, n, size=E)
    if degree == "uniform":
        tgt = rng.integers(0, n, size=E)
    elif degree == "powerlaw":
        tgt = _powerlaw_targets(rng, n, E, 2.0, min(max_degree, E))
    else:
        raise ValueError(degree)
    adj = []
    for t in range(T):
        sel = etype == t
        adj.append(np.stack([src[sel], tgt[sel]], axis=1).astype(I32))

    C = min(num_candidates, n)
    cand = np.sort(rng.choice(n, size=C, replace=False)).astype(I32)  # np.unique order (data.py:141)
    if buggy is None:
        buggy = bool(rng.integers(0, 2))
    refs = {"candidate_nodes": cand}
    if buggy and C > 0:
        loc = int(rng.integers(0, C))
        node = int(cand[loc])
        which = int(rng.integers(0, n_text + n_var + n_swap))
        refs["target_rewrite_nodes"] = np.full(n_text, node, dtype=I32)
        refs["varmisused_node_ids"] = np.full(n_var, node, dtype=I32)
        refs["candidate_symbol_node_ids"] = rng.integers(0, n, size=n_var).astype(I32)
        refs["call_node_ids"] = np.full(n_swap, node, dtype=I32)
        refs["candidate_swapped_node_ids"] = rng.integers(0, n, size=(n_swap, 2)).astype(I32)
        # original rewrite ids: text first, then var-misuse, then arg-swaps
        return BaseTensorizedBugLabGnn(
            graph_data=TensorizedGraphData(ids, lens, adj, refs),
            target_location_node_idx=loc,
            target_rewrites=rng.integers(0, rewrite_vocab_size, size=n_text).tolist(),
            target_rewrite_to_location_group=[loc] * n_text,
            correct_rewrite_target=which if which < n_text else None,
            text_rewrite_original_idx=list(range(n_text)),
            candidate_symbol_to_varmisused_node=[loc] * n_var,
            correct_candidate_symbol_node=(which - n_text) if n_text <= which < n_text + n_var else None,
            candidate_rewrite_original_idx=list(range(n_text, n_text + n_var)),
            swapped_pair_to_call=[loc] * n_swap,
            correct_swapped_pair=(which - n_text - n_var) if which >= n_text + n_var else None,
            pair_rewrite_original_idx=list(range(n_text + n_var, n_text + n_var + n_swap)),
            num_rewrite_locations_considered=C,
            rewrite_logprobs=None,
        )
    for k in ("target_rewrite_nodes", "varmisused_node_ids", "candidate_symbol_node_ids", "call_node_ids"):
        refs[k] = np.zeros(0, dtype=I32)
    refs["candidate_swapped_node_ids"] = np.zeros((0, 2), dtype=I32)
    return BaseTensorizedBugLabGnn(
        graph_data=TensorizedGraphData(ids, lens, adj, refs),
        target_location_node_idx=None,
        target_rewrites=[],
        target_rewrite_to_location_group=[],
        correct_rewrite_target=None,
        text_rewrite_original_idx=[],
        candidate_symbol_to_varmisused_node=[],
        correct_candidate_symbol_node=None,
        candidate_rewrite_original_idx=[],
        swapped_pair_to_call=[],
        correct_swapped_pair=None,
        pair_rewrite_original_idx=[],
        num_rewrite_locations_considered=C,
        rewrite_logprobs=None,
    )


def make_samples(num_graphs: int, seed: int = 0, **kw) -> List[BaseTensorizedBugLabGnn]:
    rng = np.random.default_rng(seed)
    out = []
    for b in range(num_graphs):
        kw_b = dict(kw)
        if "buggy" not in kw_b:
            kw_b["buggy"] = (b % 2 == 0)  # exactly 50 % buggy, deterministic
        out.append(make_sample(rng, **kw_b))
    return out

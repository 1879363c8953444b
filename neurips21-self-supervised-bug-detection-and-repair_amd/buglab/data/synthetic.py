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


# ------------------------------------------------------------------------------------------------
# Synthetic *raw* datapoints (the msgpack dict format, reference buglab/representations/data.py:14-20,
# 130-137) for exercising the host path: metadata pass, tensorize, rewrite bookkeeping, predict.
_KINDS = ["Module", "FunctionDef", "Call", "Name", "BinaryOperation", "Assign", "Return", "If", "Attribute"]
_WORDS = ["get", "set", "value", "index", "count", "name", "path", "file", "data", "item", "key", "node", "size", "max", "min"]


def _identifier(rng) -> str:
    k = int(rng.integers(1, 4))
    parts = [_WORDS[int(rng.integers(0, len(_WORDS)))] for _ in range(k)]
    return "_".join(parts) if rng.integers(0, 2) else parts[0] + "".join(p.capitalize() for p in parts[1:])


def make_buglab_datapoint(rng: np.random.Generator, num_syntax_nodes: int = 40, num_tokens: int = 30, buggy: bool = True,
                          package: str = "synthetic"):
    """One BugLabData dict with Child / NextToken / OccurrenceOf / Sibling edges, Call nodes with
    `args` children, and candidate rewrites of all three scout families at several locations."""
    nodes: List[str] = []
    child, sibling, next_token, occ = [], [], [], []
    for i in range(num_syntax_nodes):
        nodes.append(_KINDS[int(rng.integers(0, len(_KINDS)))] if i else "Module")
        if i:
            child.append([int(rng.integers(0, i)), i])
    tok0 = len(nodes)
    for j in range(num_tokens):
        nodes.append(_identifier(rng) if rng.integers(0, 3) else ["(", ")", "+", "=", ":"][int(rng.integers(0, 5))])
        if j:
            next_token.append([tok0 + j - 1, tok0 + j])
        child.append([int(rng.integers(0, num_syntax_nodes)), tok0 + j])
    sym0 = len(nodes)
    for s in range(5):
        nodes.append(_identifier(rng))
        for _ in range(3):
            occ.append([tok0 + int(rng.integers(0, num_tokens)), sym0 + s])
    for i in range(1, num_syntax_nodes - 1, 3):
        sibling.append([i, i + 1])
    # two Call nodes with >= 2 `args` children each
    calls = []
    for c in range(2):
        call = len(nodes)
        nodes.append("Call")
        child.append([0, call])
        args = []
        for a in range(3):
            arg = len(nodes)
            nodes.append("Name")
            child.append([call, arg, "args"])
            args.append(arg)
        calls.append((call, args))

    reference_nodes, rewrites, metadata, ranges = [], [], [], []

    def add(node, rewrite, meta):
        reference_nodes.append(int(node))
        rewrites.append(rewrite)
        metadata.append(meta)
        ranges.append(((0, 0), (0, 1)))

    op_node = tok0 + 1
    for op in ("+", "-", "*", "<="):
        add(op_node, ("ReplaceText", op), ("BinaryOperatorRewriteScout", None))
    var_node = tok0 + 3
    for s in range(4):
        add(var_node, ("ReplaceText", nodes[sym0 + s]), ("VariableMisuseRewriteScout", sym0 + s))
    for call, args in calls:
        add(call, ("ArgSwap", (0, 1)), ("ArgSwapRewriteScout", None))
        add(call, ("ArgSwap", (1, 2)), ("ArgSwapRewriteScout", None))
    add(tok0 + 5, ("ReplaceText", "True"), ("LiteralRewriteScout", None))
    target = int(rng.integers(0, len(rewrites))) if buggy else None
    return {
        "graph": {
            "nodes": nodes,
            "edges": {"Child": child, "NextToken": next_token, "OccurrenceOf": occ, "Sibling": sibling},
            "path": f"{package}/f.py",
            "text": "",
            "reference_nodes": reference_nodes,
            "code_range": ((0, 0), (1, 0)),
        },
        "candidate_rewrites": rewrites,
        "candidate_rewrite_metadata": metadata,
        "candidate_rewrite_ranges": ranges,
        "target_fix_action_idx": target,
        "package_name": package,
    }


def make_buglab_dataset(n: int, seed: int = 0):
    rng = np.random.default_rng(seed)
    return [make_buglab_datapoint(rng, num_syntax_nodes=int(rng.integers(20, 60)), num_tokens=int(rng.integers(15, 40)),
                                  buggy=(i % 2 == 0)) for i in range(n)]


# ------------------------------------------------------------------------------------------------
# Synthetic raw datapoints with an AST-shaped Child tree over a NextToken chain: what the sequence models'
# graph -> token projection (buglab.representations.tokenseq, reference seqmodel.py:441-589) walks --
# Assign / BinaryOperation / ComparisonTarget (incl. the two-token IsNot) / Call nodes, symbols with
# OccurrenceOf edges, data-flow style edges between tokens and syntax nodes, rewrites of all scout families.
def make_buglab_seq_datapoint(rng: np.random.Generator, num_statements: int = 6, buggy: bool = True, package: str = "synthetic"):
    nodes: List[str] = ["Module"]
    child: List[list] = []
    tokens: List[int] = []
    names: List[int] = []            # Name tokens
    symbols = [_identifier(rng) for _ in range(4)]

    def add(label, parent=None, edge_label=None):
        nodes.append(label)
        i = len(nodes) - 1
        if parent is not None:
            child.append([parent, i] if edge_label is None else [parent, i, edge_label])
        return i

    def tok(label, parent, edge_label=None):
        i = add(label, parent, edge_label)
        tokens.append(i)
        return i

    def name(parent, edge_label=None):
        i = tok(symbols[int(rng.integers(0, len(symbols)))], parent, edge_label)
        names.append(i)
        return i

    binops, comparisons, calls, literals = [], [], [], []
    for _ in range(num_statements):
        kind = int(rng.integers(0, 4))
        if kind == 0:  # x = a + b
            st = add("Assign", 0)
            name(st)
            tok("=", st)
            bo = add("BinaryOperation", st)
            name(bo)
            tok(["+", "-", "*"][int(rng.integers(0, 3))], bo)
            name(bo)
            binops.append(bo)
        elif kind == 1:  # if a < b :   /   if a is not b :
            st = add("If", 0)
            tok("if", st)
            ct = add("ComparisonTarget", st)
            name(ct)
            if rng.integers(0, 3) == 0:
                two = add("IsNot", ct)
                tok("is", two)
                tok("not", two)
            else:
                tok(["<", "<=", "==", "!="][int(rng.integers(0, 4))], ct)
            name(ct)
            tok(":", st)
            comparisons.append(ct)
        elif kind == 2:  # f ( a , b , c )
            st = add("Expr", 0)
            call = add("Call", st)
            name(call)
            tok("(", call)
            args = []
            for a in range(3):
                args.append(name(call, "args"))
                if a < 2:
                    tok(",", call)
            tok(")", call)
            calls.append(call)
        else:  # x += 1
            st = add("AugAssign", 0)
            name(st)
            tok("+=", st)
            literals.append(tok(str(int(rng.integers(0, 3))), st))
    if not names:
        st = add("Expr", 0)
        name(st)
    next_token = [[tokens[i], tokens[i + 1]] for i in range(len(tokens) - 1)]
    sym_nodes, occ = {}, []
    for t in names:
        s = nodes[t]
        if s not in sym_nodes:
            nodes.append(s)
            sym_nodes[s] = len(nodes) - 1
        occ.append([t, sym_nodes[s]])
    pick = lambda pool: pool[int(rng.integers(0, len(pool)))]
    syntax = [i for i in range(len(nodes)) if i not in tokens and i not in sym_nodes.values()]
    edges = {
        "Child": child, "NextToken": next_token, "OccurrenceOf": occ,
        "Sibling": [[child[i][1], child[i + 1][1]] for i in range(0, len(child) - 1, 3)],
        "LastMayWrite": [[pick(names), pick(names)] for _ in range(max(1, len(names) // 2))],
        "NextMayUse": [[pick(names), pick(names)] for _ in range(max(1, len(names) // 2))],
        "ComputedFrom": [[pick(syntax), pick(names)] for _ in range(3)] + [[pick(names), pick(syntax)] for _ in range(2)],
    }
    reference_nodes, rewrites, metadata, ranges = [], [], [], []

    def rw(node, rewrite, meta):
        reference_nodes.append(int(node))
        rewrites.append(rewrite)
        metadata.append(meta)
        ranges.append(((0, 0), (0, 1)))

    for bo in binops[:2]:
        for op in ("+", "-", "*", "/"):
            rw(bo, ("ReplaceText", op), ("BinaryOperatorRewriteScout", None))
    for ct in comparisons[:2]:
        for op in ("<", "<=", "=="):
            rw(ct, ("ReplaceText", op), ("ComparisonOperatorRewriteScout", None))
    for t in names[:3]:
        for s in sym_nodes.values():
            rw(t, ("ReplaceText", nodes[s]), ("VariableMisuseRewriteScout", int(s)))
    for call in calls[:2]:
        rw(call, ("ArgSwap", (0, 1)), ("ArgSwapRewriteScout", None))
        rw(call, ("ArgSwap", (1, 2)), ("ArgSwapRewriteScout", None))
    for lit in literals[:1]:
        rw(lit, ("ReplaceText", "1"), ("LiteralRewriteScout", None))
    target = int(rng.integers(0, len(rewrites))) if (buggy and rewrites) else None
    return {
        "graph": {"nodes": nodes, "edges": edges, "path": f"{package}/seq.py", "text": "", "reference_nodes": reference_nodes,
                  "code_range": ((0, 0), (1, 0))},
        "candidate_rewrites": rewrites, "candidate_rewrite_metadata": metadata, "candidate_rewrite_ranges": ranges,
        "target_fix_action_idx": target, "package_name": package,
    }


def make_buglab_seq_dataset(n: int, seed: int = 0, min_statements: int = 4, max_statements: int = 9):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        d = make_buglab_seq_datapoint(rng, num_statements=int(rng.integers(min_statements, max_statements + 1)), buggy=(len(out) % 2 == 0))
        if d["candidate_rewrites"]:
            out.append(d)
    return out

"""Graph -> token-sequence projection of the sequence models (`seq-great`, `seq-rat`, ...): every node of a BugLab
program graph is mapped to a position in the token sequence, and the edges that are kept are re-targeted to token
positions.  Counterpart of `SeqBugLabModel.__to_token_data` / `__extract_token_sequence`
(reference buglab/models/seqmodel.py:441-617); `tests/test_seq_host_golden.py` pins it to the reference's own
function on synthetic graphs that exercise every rule.

Rules (the reference's, by node label; evaluated against the mapping built SO FAR -- it grows while the Child edges
are walked in order, each edge's parent first, then its child):
  token                          -> its position
  ComparisonTarget               -> its first child that is a comparison-operator token; `IsNot` / `NotIn` children
                                    stand for two tokens: the first token found below them
  BinaryOperation                -> its operator child if that one is mapped, else whatever the node's parent maps to
  Assign / AugAssign             -> the first child whose label contains "="
  anything else                  -> the first already-mapped child, else depth-first into the unmapped children
                                    (last one first); when nothing is found, whatever the node's parent maps to
  symbol (target of OccurrenceOf) -> the earliest position among its occurrences
A graph the rules cannot resolve is rejected (the caller drops the sample), like the exceptions the reference
catches in `tensorize` (:637-641).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

# edge kinds that do not become token-to-token relations (reference seqmodel.py:449-460)
NON_TOKEN_EDGES = frozenset({"NextToken", "PossibleType", "CandidateCall", "CandidateCallDoc", "MayFormalName", "Child", "Sibling",
                             "OccurrenceOf"})
COMPARISON_TOKENS = frozenset({"<", "<=", "==", "!=", ">", ">=", "is", "in", "not"})
TWO_TOKEN_COMPARISONS = frozenset({"IsNot", "NotIn"})
BINARY_OPERATOR_TOKENS = frozenset({"+", "-", "*", "/", "//", "**", "%", "@", ">>", "<<", "|", "&", "^"})


class TokenProjectionError(Exception):
    """The graph has no single connected token chain, or a node cannot be mapped to a token."""


def extract_token_sequence(graph) -> List[int]:
    """Node ids of the tokens in source order, following the NextToken chain from its unique head
    (reference :591-617).  Raises TokenProjectionError for a forked, cyclic or broken chain."""
    nxt = {f: t for f, t in graph["edges"]["NextToken"]}
    heads = set(nxt.keys()) - set(nxt.values())
    if len(heads) != 1:
        raise TokenProjectionError("the tokens are not connected in one chain")
    cur = next(iter(heads))
    seq, seen = [cur], {cur}
    while cur in nxt:
        cur = nxt[cur]
        if cur in seen:
            raise TokenProjectionError("cyclic token sequence")
        seen.add(cur)
        seq.append(cur)
    if len(seq) != len(nxt) + 1:
        raise TokenProjectionError("broken token sequence")
    return seq


class _Projector:
    def __init__(self, graph, token_sequence: List[int]):
        self.labels = graph["nodes"]
        self.token_set = set(token_sequence)
        self.pos: Dict[int, int] = {t: i for i, t in enumerate(token_sequence)}  # node id -> token position (grows)
        self.children: Dict[int, List[int]] = {}
        for e in graph["edges"]["Child"]:
            self.children.setdefault(e[0], []).append(e[1])

    def kids(self, n: int) -> List[int]:
        return self.children.get(n, ())

    def first_token_below(self, n: int) -> int:
        stack = [n]
        while stack:
            cur = stack.pop()
            if cur in self.token_set:
                return cur
            stack.extend(self.kids(cur))
        raise TokenProjectionError("no token below a two-token comparison node")

    def via_parent(self, n: int) -> int:
        for parent, kids in self.children.items():  # first parent in order of first appearance in the Child list
            if n in kids:
                return self.position_of(parent)
        raise TokenProjectionError(f"node {n} has no parent to fall back to")

    def _mapped(self, n: int) -> int:
        try:
            return self.pos[n]
        except KeyError:
            raise TokenProjectionError(f"node {n} ({self.labels[n]!r}) is not mapped to a token yet") from None

    def position_of(self, n: int) -> int:
        if n in self.pos:
            return self.pos[n]
        stack = [n]
        while stack:
            cur = stack.pop()
            label = self.labels[cur]
            if label == "ComparisonTarget":
                for c in self.kids(cur):
                    if self.labels[c] in COMPARISON_TOKENS:
                        return self._mapped(c)
                    if self.labels[c] in TWO_TOKEN_COMPARISONS:
                        return self._mapped(self.first_token_below(c))
                raise TokenProjectionError("a ComparisonTarget without a comparison operator child")
            if label == "BinaryOperation":
                for c in self.kids(cur):
                    if self.labels[c] in BINARY_OPERATOR_TOKENS:
                        return self.pos[c] if c in self.pos else self.via_parent(n)
                raise TokenProjectionError("a BinaryOperation without an operator child")
            if label in ("Assign", "AugAssign"):
                for c in self.kids(cur):
                    if "=" in self.labels[c]:
                        return self._mapped(c)
                raise TokenProjectionError("an assignment without an equals child")
            for c in self.kids(cur):
                if c in self.pos:
                    return self.pos[c]
                stack.append(c)
        return self.via_parent(n)  # rarely needed (e.g. f-strings)


def project_graph_to_tokens(graph) -> Tuple[List[str], Dict[int, int], Dict[str, List[Tuple[int, int]]], List[int]]:
    """-> (token strings in order, {graph node id: token position}, {edge kind: [(from position, to position)]},
    positions of graph["reference_nodes"]).  Raises TokenProjectionError (or KeyError / ValueError on malformed input)."""
    tokens = extract_token_sequence(graph)
    pr = _Projector(graph, tokens)
    for e in graph["edges"]["Child"]:  # parent first, then child; later lookups see these entries
        pr.pos[e[0]] = pr.position_of(e[0])
        pr.pos[e[1]] = pr.position_of(e[1])
    occurrences: Dict[int, List[int]] = {}
    for f, t in graph["edges"]["OccurrenceOf"]:
        occurrences.setdefault(t, []).append(f)
    for symbol, occ in occurrences.items():
        pr.pos[symbol] = min(pr._mapped(o) for o in occ)
    edges: Dict[str, List[Tuple[int, int]]] = {}
    for kind, adj in graph["edges"].items():
        if kind in NON_TOKEN_EDGES:
            continue
        mapped = []
        for pair in adj:
            f, t = pair  # exactly two entries, like the reference's unpacking (:577)
            mapped.append((pr.position_of(f), pr.position_of(t)))
        edges[kind] = mapped
    ref_positions = []
    for ref in graph["reference_nodes"]:
        p = pr.position_of(ref)
        pr.pos[ref] = p
        ref_positions.append(p)
    return [graph["nodes"][i] for i in tokens], pr.pos, edges, ref_positions

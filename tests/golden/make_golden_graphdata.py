#!/usr/bin/env python
"""Row C1 at SIZE: `BugLabData.as_graph_data` + `add_open_vocab_nodes_and_edges` of the REFERENCE
(buglab/representations/data.py:97-167) on graphs of 3 000 / 6 000 / 20 000 nodes, where the order in which the
reference walks its `set` of token nodes (data.py:109) is no longer the sorted order -- that order numbers the subtoken
nodes and orders the HasSubtoken edges.

    python tests/golden/make_golden_graphdata.py    # rewrites reference_shard_large.msgpack.l.gz + graphdata_large.json

The shard is written by the reference's own `save_msgpack_l_gz`; the JSON holds, per datapoint, what the reference's
`as_graph_data` made of what its `load_msgpack_l_gz` read back: node count, sha256 of the node strings, per edge kind the
shape and sha256 of the int32 edge array, the candidate nodes, and the first / last HasSubtoken edges in clear.  Token
nodes sit at scattered node ids (as in real extracted graphs, where syntax nodes, tokens and symbols interleave), and the
NextToken chain visits them in a shuffled order, so neither sorted ids nor insertion order reproduce the set order."""
import hashlib
import json
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, OUT)
import make_golden as MG  # noqa: E402

_WORDS = ["value", "index", "node", "item", "count", "result", "data", "key", "name", "size", "total", "buffer", "cursor",
          "offset", "parent", "child", "left", "right", "queue", "stack", "path", "config", "http", "request", "parse"]


def _identifier(rng):
    k = int(rng.integers(1, 4))
    parts = [_WORDS[int(rng.integers(0, len(_WORDS)))] + (str(int(rng.integers(0, 40))) if rng.integers(0, 3) == 0 else "")
             for _ in range(k)]
    return "_".join(parts) if rng.integers(0, 2) else parts[0] + "".join(p.capitalize() for p in parts[1:])


def big_datapoint(rng, num_nodes, token_share):
    is_token = rng.uniform(size=num_nodes) < token_share
    is_token[0] = False
    nodes = []
    for i in range(num_nodes):
        if is_token[i]:
            nodes.append(_identifier(rng) if rng.integers(0, 4) else ["(", ")", "+", "=", ":", "1", "'s'"][int(rng.integers(0, 7))])
        else:
            nodes.append(["Module", "Call", "Name", "Assign", "If", "BinaryOperation", "Attribute"][int(rng.integers(0, 7))] if i else "Module")
    tokens = np.flatnonzero(is_token)
    chain = tokens.copy()
    # mostly in id order with shuffled stretches: NextToken follows the token stream, ids follow the traversal
    for s in range(0, len(chain) - 64, 97):
        rng.shuffle(chain[s:s + 64])
    next_token = [[int(a), int(b)] for a, b in zip(chain[:-1], chain[1:])]
    child = [[int(rng.integers(0, i)), i] for i in range(1, num_nodes)]
    syntax = np.flatnonzero(~is_token)
    sibling = [[int(a), int(b)] for a, b in zip(syntax[1:-1:3], syntax[2::3])]
    reference_nodes = [int(x) for x in rng.choice(tokens, size=12, replace=True)]
    rewrites = [("ReplaceText", "+")] * 12
    metadata = [("BinaryOperatorRewriteScout", None)] * 12
    return {
        "graph": {"nodes": nodes, "edges": {"Child": child, "NextToken": next_token, "Sibling": sibling}, "path": "big/f.py", "text": "",
                  "reference_nodes": reference_nodes, "code_range": ((0, 0), (1, 0))},
        "candidate_rewrites": rewrites, "candidate_rewrite_metadata": metadata,
        "candidate_rewrite_ranges": [((0, 0), (0, 1))] * 12, "target_fix_action_idx": 3, "package_name": "big",
    }


def digest(graph_data, target_node_idx):
    """(shared with tests/test_graphdata_golden.py) what is compared: every node string and every edge, in order."""
    nodes = list(graph_data.node_information)
    rec = {"num_nodes": len(nodes), "nodes_sha256": hashlib.sha256("\x00".join(nodes).encode("utf-8")).hexdigest(),
           "target_node_idx": None if target_node_idx is None else int(target_node_idx),
           "candidate_nodes": [int(x) for x in np.asarray(graph_data.reference_nodes["candidate_nodes"]).reshape(-1)], "edges": {}}
    for kind, arr in graph_data.edges.items():
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.int32).reshape(-1, 2))
        rec["edges"][kind] = {"count": int(a.shape[0]), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
    hs = np.asarray(graph_data.edges["HasSubtoken"], dtype=np.int32).reshape(-1, 2)
    rec["has_subtoken_head"] = hs[:24].tolist()
    rec["has_subtoken_tail"] = hs[-24:].tolist()
    return rec


def main():
    MG._install_stubs()
    sys.path.insert(1, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "_amd_vocabulary", os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd", "buglab/runtime/vocabulary.py"))
    vocab_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vocab_mod)
    sys.modules["dpu_utils.codeutils"].split_identifier_into_parts = vocab_mod.split_identifier_into_parts  # dpu_utils is absent
    sys.modules["dpu_utils.mlutils"].Vocabulary = type("Vocabulary", (), {"get_pad": staticmethod(lambda: "%PAD%")})
    sys.modules["ptgnn.neuralmodels.gnn"].GraphData = type("GraphData", (), {"__init__": lambda self, **kw: self.__dict__.update(kw)})
    sys.path.insert(0, "/root/reference")
    from buglab.representations.data import BugLabData  # noqa: reference code
    from buglab.utils.msgpackutils import load_msgpack_l_gz, save_msgpack_l_gz  # noqa: reference code

    assert sys.modules["buglab"].__file__.startswith("/root/reference")
    rng = np.random.default_rng(2024)
    points = [big_datapoint(rng, 3000, 0.35), big_datapoint(rng, 6000, 0.18), big_datapoint(rng, 20000, 0.22)]  # token counts just below a set-resize threshold: ids wrap around the table
    shard = os.path.join(OUT, "reference_shard_large.msgpack.l.gz")
    save_msgpack_l_gz(points, shard)
    out = {"python": sys.version.split()[0], "datapoints": []}
    differs = 0
    for d in load_msgpack_l_gz(shard):
        toks = set()
        for a, b in d["graph"]["edges"]["NextToken"]:
            toks.add(a)
            toks.add(b)
        differs += list(toks) != sorted(toks)
        gd, tgt = BugLabData.as_graph_data(d)
        out["datapoints"].append(digest(gd, tgt))
    assert differs == len(points), "the fixture must exercise set order != sorted order"
    with open(os.path.join(OUT, "graphdata_large.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"wrote {shard} ({os.path.getsize(shard)} bytes) and graphdata_large.json; set order differs from sorted order in {differs}/{len(points)}")


if __name__ == "__main__":
    main()

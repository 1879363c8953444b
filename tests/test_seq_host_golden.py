"""CPU: host side of the sequence models (`seq-great`) against fixtures produced by the REFERENCE's own
`SeqBugLabModel` (tests/golden/make_golden_seq.py; reference seqmodel.py:441-975): the graph -> token projection,
`tensorize`, and the minibatch index arrays.  Bit-exact on every index."""
import copy
import gzip
import json
import os
from pathlib import Path

import numpy as np
import pytest

from buglab.representations.tokenseq import project_graph_to_tokens


@pytest.fixture(scope="module")
def fx(golden_dir):
    with gzip.open(os.path.join(golden_dir, "seq_host.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


@pytest.fixture(scope="module")
def model(fx):
    from buglab.models.modelregistry import load_model

    m = load_model({"modelName": "seq-great", "hidden_state_size": 32, "num_layers": 1, "num_heads": 4, "intermediate_dimension_size": 48,
                    "max_seq_size": fx["max_seq_size"]}, Path("/tmp/_bl_seq_golden.pkl.gz"))[0]
    m.compute_metadata(copy.deepcopy(fx["datapoints"]))
    return m


def test_token_projection_matches_reference(fx):
    n_bad = 0
    for d, want in zip(fx["datapoints"], fx["token_data"]):
        if want is None:
            with pytest.raises(Exception):
                project_graph_to_tokens(copy.deepcopy(d["graph"]))
            n_bad += 1
            continue
        tokens, mapping, edges, refs = project_graph_to_tokens(copy.deepcopy(d["graph"]))
        assert tokens == want["tokens"]
        assert sorted([int(k), int(v)] for k, v in mapping.items()) == want["mapping"]
        assert {k: [[int(a), int(b)] for a, b in v] for k, v in edges.items()} == want["edges"]
        assert [int(r) for r in refs] == want["reference_positions"]
    assert n_bad == 1  # the forked token chain is rejected by both


def test_metadata_pass(fx, model):
    assert model.edge_types == fx["edge_types"]
    assert model._target_rewrite_ops.token_to_id == fx["operator_vocabulary"]


def _points(fx, mode):
    return [d for d in fx["datapoints"] if mode == "all" or "candidate_rewrite_logprobs" not in d]


def _tensorize(model, points, mode):
    if mode == "all":
        with model._tensorize_all_location_rewrites():
            return [model.tensorize(copy.deepcopy(d)) for d in points]
    return [model.tensorize(copy.deepcopy(d)) for d in points]


L = lambda a: np.asarray(a).astype(np.int64).tolist()


@pytest.mark.parametrize("mode", ["train", "all"])
def test_tensorize_matches_reference(fx, model, mode):
    ours = _tensorize(model, _points(fx, mode), mode)
    for t, want in zip(ours, fx["modes"][mode]["tensorized"]):
        if want is None:
            assert t is None
            continue
        b, refs = t.base, t.base.graph_data.reference_nodes
        ours_edges = {k: L(a) for k, a in zip(model.edge_types, b.graph_data.adjacency_lists) if len(a)}
        assert ours_edges == {k: v for k, v in (tuple(kv) for kv in want["intra_token_edges"]) if len(v)}
        assert L(refs["candidate_nodes"]) == want["candidate_location_idxs"]
        assert b.target_location_node_idx == want["target_location_idx"]
        assert L(refs["target_rewrite_nodes"]) == want["target_rewrite_node_ids"]
        assert L(refs["varmisused_node_ids"]) == want["varmisused_node_ids"]
        assert L(refs["candidate_symbol_node_ids"]) == want["candidate_symbol_node_ids"]
        assert L(refs["call_node_ids"]) == want["call_node_ids"]
        assert L(refs["candidate_swapped_node_ids"]) == [list(p) for p in want["candidate_swapped_node_ids"]]
        for ours_f, ref_f in (("target_rewrites", "target_rewrites"), ("target_rewrite_to_location_group", "target_rewrite_to_location_group"),
                              ("text_rewrite_original_idx", "text_rewrite_original_idx"),
                              ("candidate_symbol_to_varmisused_node", "candidate_symbol_to_varmisused_node"),
                              ("candidate_rewrite_original_idx", "candidate_rewrite_original_idx"), ("swapped_pair_to_call", "swapped_pair_to_call"),
                              ("pair_rewrite_original_idx", "pair_rewrite_original_idx")):
            assert L(getattr(b, ours_f)) == want[ref_f], ours_f
        for f in ("correct_rewrite_target", "correct_candidate_symbol_node", "correct_swapped_pair", "num_rewrite_locations_considered"):
            assert getattr(b, f) == want[f], f
        if want["rewrite_logprobs"] is None:
            assert b.rewrite_logprobs is None
        else:
            assert np.array_equal(np.asarray(b.rewrite_logprobs, dtype=np.float64), np.asarray(want["rewrite_logprobs"], dtype=np.float64))
        assert [model.token_embedder.vocabulary.get_name_for_id(int(i)) is not None for i in b.graph_data.token_ids[:, 0]]


@pytest.mark.parametrize("mode", ["train", "all"])
def test_padded_minibatch_matches_reference_finalize(fx, model, mode):
    """Our [B * L] node layout vs the reference's (sample, position) pairs (seqmodel.py:770-975)."""
    from buglab.models.seqmodel import collate_sequences

    samples = [t for t in _tensorize(model, _points(fx, mode), mode) if t is not None]
    mb = collate_sequences(samples, len(model.edge_types))
    ref = fx["modes"][mode]["minibatch"]
    gd = mb["graph_data"]
    B, Lp = gd["seq_batch"], gd["seq_len"]
    assert gd["seq_lens"].tolist() == ref["token_sequence_lengths"] and Lp % 4 == 0 and Lp >= max(ref["token_sequence_lengths"])
    pairs = lambda flat: [[int(f) // Lp, int(f) % Lp] for f in np.asarray(flat).reshape(-1)]
    r = gd["reference_node_ids"]
    assert pairs(r["candidate_nodes"]) == ref["candidate_location_idxs"]
    assert pairs(r["target_rewrite_nodes"]) == ref["target_rewrite_node_ids"]
    assert pairs(r["varmisused_node_ids"]) == ref["varmisused_node_ids"]
    assert pairs(r["candidate_symbol_node_ids"]) == ref["candidate_symbol_node_ids"]
    assert pairs(r["call_node_ids"]) == ref["call_node_ids"]
    sw = np.asarray(r["candidate_swapped_node_ids"]).reshape(-1, 2)
    assert [[int(a) // Lp, int(a) % Lp, int(b) % Lp] for a, b in sw] == ref["candidate_swapped_node_ids"]
    assert all(int(a) // Lp == int(b) // Lp for a, b in sw)
    has_bug = np.asarray(mb["has_bug"]).astype(bool)
    assert has_bug.tolist() == ref["has_bug"]
    ours_t = np.asarray(mb["correct_candidate_node_idxs"])
    assert ours_t[has_bug].tolist() == np.asarray(ref["target_location_idxs"])[has_bug].tolist()  # unused for bug-free samples
    for k in ("target_rewrites", "rewrite_to_location_group", "correct_rewrite_idxs", "candidate_symbol_to_location_group",
              "correct_candidate_symbols", "swapped_pair_to_call_location_group", "correct_swapped_pair", "text_rewrite_idxs",
              "candidate_rewrite_idxs", "pair_rewrite_idxs", "rewrite_to_graph_id"):
        assert L(mb[k]) == ref[k], k
    for k in ("text_rewrite_original_idxs", "candidate_rewrite_original_idxs", "pair_rewrite_original_idx"):
        assert [list(map(int, x)) for x in mb[k]] == ref[k], k
    if "rewrite_logprobs" in ref:
        assert np.array_equal(np.asarray(mb["rewrite_logprobs"], dtype=np.float32), np.asarray(ref["rewrite_logprobs"], dtype=np.float32))
    else:
        assert mb.get("rewrite_logprobs") is None
    # edges: the query-row CSR holds every (sample, source, target, type) once per direction
    rp, key, code = gd["erow_ptr"], gd["ekey"], gd["ecode"]
    rows = np.repeat(np.arange(B * Lp), np.diff(rp))
    fwd = code % 2 == 0
    ours_e = sorted(zip((rows[fwd] // Lp).tolist(), (rows[fwd] % Lp).tolist(), key[fwd].tolist(), (code[fwd] // 2).tolist()))
    want_e = sorted((int(s), int(a), int(b), int(t)) for (s, a, b), t in zip(ref["edges"], ref["edge_types"]))
    assert ours_e == want_e
    rev = ~fwd
    ours_r = sorted(zip((rows[rev] // Lp).tolist(), key[rev].tolist(), (rows[rev] % Lp).tolist(), (code[rev] // 2).tolist()))
    assert ours_r == want_e
    # token strings of every sequence, position by position
    vocab = model.token_embedder.vocabulary
    for b, toks in enumerate(ref["input_tokens"]):
        assert gd["token_lens"][b * Lp : b * Lp + len(toks)].min() >= 1
        assert (gd["token_ids"][b * Lp + len(toks) : (b + 1) * Lp] == 0).all()

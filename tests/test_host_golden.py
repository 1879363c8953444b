"""CPU: the host-side integer work against fixtures produced by the
REFERENCE's own `_compute_rewrite_data`, `tensorize`, `extend/finalize_minibatch` and
`_iter_per_sample_results` (tests/golden/make_golden_host.py; reference basemodel.py:80-346,
gnn.py:361-604).  Bit-exact: every index list must be identical, float32 log-probabilities equal."""
import copy
import gzip
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from buglab.data import collate as C


@pytest.fixture(scope="module")
def fx(golden_dir):
    with gzip.open(os.path.join(golden_dir, "host_rewrite_data.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


@pytest.fixture(scope="module")
def model(fx):
    from buglab.models.modelregistry import load_model

    m = load_model({"modelName": "gnn-mlp", "hidden_state_size": 32, "num_layers": 4}, Path("/tmp/_bl_host_golden.pkl.gz"))[0]
    m.compute_metadata(copy.deepcopy(fx["datapoints"]))
    return m


def _undict(pairs):
    return {k: v for k, v in pairs}


def test_operator_vocabulary_is_the_sorted_one(fx, model):
    assert model._target_rewrite_ops.token_to_id == fx["operator_vocabulary"]


def _points(fx, mode):
    return [d for d in fx["datapoints"] if mode == "all" or "candidate_rewrite_logprobs" not in d]


def _jsonify(x):
    if isinstance(x, dict):
        return [[_jsonify(k), _jsonify(v)] for k, v in x.items()]
    if isinstance(x, (list, tuple)):
        return [_jsonify(v) for v in x]
    if isinstance(x, np.ndarray):
        return _jsonify(x.tolist())
    if isinstance(x, np.integer):
        return int(x)
    return x


@pytest.mark.parametrize("mode", ["train", "all"])
def test_compute_rewrite_data_and_tensorize_match_reference(fx, model, mode):
    from buglab.representations.data import BugLabData

    ref = fx["modes"][mode]
    ctx = model._tensorize_all_location_rewrites() if mode == "all" else None
    if ctx:
        ctx.__enter__()
    try:
        for d, want_rw, want_t, want_cand in zip(_points(fx, mode), ref["rewrite_data"], ref["tensorized"], ref["candidate_nodes"]):
            gd, _ = BugLabData.as_graph_data(copy.deepcopy(d))
            cand = gd.reference_nodes["candidate_nodes"]
            assert np.asarray(cand).tolist() == want_cand
            got = model._compute_rewrite_data(copy.deepcopy(d), cand)
            assert _jsonify(got) == want_rw
            t = model.tensorize(copy.deepcopy(d))
            for f, want in want_t.items():
                have = getattr(t, f)
                if f == "rewrite_logprobs" and want is not None:
                    assert np.array_equal(np.asarray(have, dtype=np.float64), np.asarray(want, dtype=np.float64))
                else:
                    assert _jsonify(have) == want, f
    finally:
        if ctx:
            ctx.__exit__(None, None, None)


def _tensorize_all(model, points, mode):
    if mode == "all":
        with model._tensorize_all_location_rewrites():
            return [model.tensorize(copy.deepcopy(d)) for d in points]
    return [model.tensorize(copy.deepcopy(d)) for d in points]


@pytest.mark.parametrize("mode", ["train", "all"])
def test_collated_minibatch_matches_reference_extend_finalize(fx, model, mode):
    """The vectorised collator vs the reference's element-by-element `extend_minibatch_with` +
    `finalize_minibatch` (gnn.py:463-604) on the same tensorised samples."""
    samples = _tensorize_all(model, _points(fx, mode), mode)
    mb = C.collate_samples(samples, model.gnn_model.num_presented_edge_types)
    ref = fx["modes"][mode]["minibatch"]
    for k, want in ref.items():
        if k in ("num_nodes_per_graph", "candidate_node_ids"):
            continue
        if k in ("text_rewrite_original_idxs", "candidate_rewrite_original_idxs", "pair_rewrite_original_idx"):
            assert [list(map(int, x)) for x in mb[k]] == want, k
        elif k == "rewrite_logprobs":
            assert np.array_equal(np.asarray(mb[k], dtype=np.float32), np.asarray(want, dtype=np.float32)), k
        elif k == "has_bug":
            assert np.asarray(mb[k]).astype(bool).tolist() == want
        else:
            assert np.asarray(mb[k]).astype(np.int64).tolist() == want, k
    if "rewrite_logprobs" not in ref:
        assert mb.get("rewrite_logprobs") is None
    gd = mb["graph_data"]
    assert np.bincount(gd["node_to_graph"], minlength=len(samples)).tolist() == ref["num_nodes_per_graph"]
    assert gd["reference_node_ids"]["candidate_nodes"].tolist() == ref["candidate_node_ids"]


@pytest.mark.parametrize("mapped", [False, True])
def test_unbatching_matches_reference(fx, model, mapped):
    """`_iter_per_sample_results` (reference basemodel.py:240-346) on the reference's minibatch and seeded
    float32 log-probabilities, without and with a many-to-one node mapping."""
    points = _points(fx, "all")
    samples = _tensorize_all(model, points, "all")
    mb = C.collate_samples(samples, model.gnn_model.num_presented_edge_types)
    u = fx["unbatch"]
    f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
    mbt = {k: (torch.as_tensor(np.asarray(v)) if isinstance(v, np.ndarray) else v) for k, v in mb.items()}
    maps = [{int(n): int(n) // 2 for n in range(len(d["graph"]["nodes"]) + 64)} for d in points] if mapped else None
    res = list(model._iter_per_sample_results(mbt, np.asarray(u["sample_idx"]), np.asarray(u["loc_logprobs"], dtype=np.float32),
                                              f32(u["swap_logprobs"]), len(points), points, f32(u["text_logprobs"]),
                                              f32(u["var_logprobs"]), node_mappings=maps))
    want = u["results_mapped" if mapped else "results"]
    assert len(res) == len(want)
    for (point, loc, rw), w in zip(res, want):
        assert {int(k): float(v) for k, v in loc.items()} == {int(k): float(v) for k, v in w["location_logprobs"]}
        assert [float(x) for x in rw] == w["rewrite_logprobs"]

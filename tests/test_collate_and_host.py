"""CPU: collator, rewrite bookkeeping, registry / model API, persistence -- the host half of the
drop-in boundary."""
import copy
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from buglab.data import collate as C
from buglab.data.synthetic import make_buglab_dataset, make_samples


def _reference_style_minibatch(samples):
    """Element-by-element re-enactment of the reference's extend_minibatch_with offsets
    (buglab/models/gnn.py:463-542), independent of the vectorised collator."""
    mb = {k: [] for k in ("has_bug", "correct_candidate_node_idxs", "target_rewrites", "rewrite_to_location_group", "correct_rewrite_idxs",
                          "text_rewrite_idxs", "candidate_symbol_to_location_group", "correct_candidate_symbols", "candidate_rewrite_idxs",
                          "swapped_pair_to_call_location_group", "correct_swapped_pair", "pair_rewrite_idxs", "rewrite_to_graph_id")}
    n_target_nodes = n_rewrites = n_groups = 0
    for g, s in enumerate(samples):
        if s.target_location_node_idx is None:
            mb["has_bug"].append(False); mb["correct_candidate_node_idxs"].append(0)
        else:
            mb["has_bug"].append(True); mb["correct_candidate_node_idxs"].append(s.target_location_node_idx + n_target_nodes)
        n_target_nodes += len(s.graph_data.reference_nodes["candidate_nodes"])
        in_dp = 0
        if s.correct_rewrite_target is not None:
            mb["correct_rewrite_idxs"].append(s.correct_rewrite_target + len(mb["target_rewrites"]))
        mb["target_rewrites"].extend(s.target_rewrites)
        mb["rewrite_to_location_group"].extend(t + n_groups for t in s.target_rewrite_to_location_group)
        mb["text_rewrite_idxs"].extend(t + n_rewrites for t in s.text_rewrite_original_idx)
        in_dp += len(s.text_rewrite_original_idx)
        if s.correct_candidate_symbol_node is not None:
            mb["correct_candidate_symbols"].append(s.correct_candidate_symbol_node + len(mb["candidate_symbol_to_location_group"]))
        mb["candidate_symbol_to_location_group"].extend(t + n_groups for t in s.candidate_symbol_to_varmisused_node)
        mb["candidate_rewrite_idxs"].extend(t + n_rewrites for t in s.candidate_rewrite_original_idx)
        in_dp += len(s.candidate_rewrite_original_idx)
        if s.correct_swapped_pair is not None:
            mb["correct_swapped_pair"].append(s.correct_swapped_pair + len(mb["swapped_pair_to_call_location_group"]))
        mb["swapped_pair_to_call_location_group"].extend(t + n_groups for t in s.swapped_pair_to_call)
        mb["pair_rewrite_idxs"].extend(t + n_rewrites for t in s.pair_rewrite_original_idx)
        in_dp += len(s.pair_rewrite_original_idx)
        n_rewrites += in_dp
        n_groups += s.num_rewrite_locations_considered
        mb["rewrite_to_graph_id"].extend([g] * in_dp)
    return mb


def test_collate_matches_reference_offsets_and_csr_invariants():
    samples = make_samples(7, seed=3, num_nodes=50, num_messages=230, num_edge_types=6, vocab_size=99, num_candidates=7)
    mb = C.collate_samples(samples, 6)
    ref = _reference_style_minibatch(samples)
    for k, v in ref.items():
        np.testing.assert_array_equal(np.asarray(mb[k]).astype(np.int64), np.asarray(v, dtype=np.int64), err_msg=k)
    gd = mb["graph_data"]
    N, E = gd["token_ids"].shape[0], gd["msg_src"].shape[0]
    assert N == 7 * 50 and E == 7 * 230 and gd["type_ptr"][-1] == E
    for t in range(6):  # type-major, target-sorted inside a type
        seg = gd["msg_tgt"][gd["type_ptr"][t]:gd["type_ptr"][t + 1]]
        assert (np.diff(seg) >= 0).all()
    # the multiset of (type, src, tgt) triples is preserved
    want = sorted((t, int(a[0]) + 50 * b, int(a[1]) + 50 * b) for b, s in enumerate(samples) for t, adj in enumerate(s.graph_data.adjacency_lists) for a in adj)
    types = np.repeat(np.arange(6), np.diff(gd["type_ptr"]))
    got = sorted(zip(types.tolist(), gd["msg_src"].tolist(), gd["msg_tgt"].tolist()))
    assert want == got
    for ptr, items, key in ((gd["tgt_ptr"], gd["tgt_msgs"], gd["msg_tgt"]), (gd["src_ptr"], gd["src_msgs"], gd["msg_src"])):
        assert ptr[0] == 0 and ptr[-1] == E and sorted(items.tolist()) == list(range(E))
        for n in (0, 17, N - 1):
            seg = items[ptr[n]:ptr[n + 1]]
            assert (key[seg] == n).all() and (np.diff(seg) > 0).all()  # ascending ids -> lowest-id tie rule
    # every graph's nodes stay inside the graph
    g_of = gd["node_to_graph"]
    assert (g_of[gd["msg_src"]] == g_of[gd["msg_tgt"]]).all()
    assert gd["loc_group_ptr"][-1] == gd["reference_node_ids"]["candidate_nodes"].shape[0] + 7


def test_to_device_single_blob_roundtrip():
    mb = C.collate_samples(make_samples(3, seed=1, num_nodes=30, num_messages=90, num_edge_types=4, vocab_size=50, num_candidates=5), 4)
    d = C.to_device(mb, "cpu")
    np.testing.assert_array_equal(d["graph_data"]["msg_src"].numpy(), mb["graph_data"]["msg_src"])
    np.testing.assert_array_equal(d["graph_data"]["reference_node_ids"]["candidate_swapped_node_ids"].numpy(), mb["graph_data"]["reference_node_ids"]["candidate_swapped_node_ids"])
    assert d["has_bug"].dtype == torch.bool and d["graph_data"]["msg_src"].dtype == torch.int32
    assert d["graph_data"]["msg_src"].data_ptr() % 16 == 0 and d["graph_data"]["tgt_ptr"].data_ptr() % 16 == 0
    # all index tensors are views of ONE allocation
    base = d["graph_data"]["_blob"]
    lo, hi = base.data_ptr(), base.data_ptr() + base.numel() * 4
    assert lo <= d["target_rewrites"].data_ptr() < hi and lo <= d["graph_data"]["src_msgs"].data_ptr() < hi


def test_empty_and_ragged_minibatches():
    from buglab.data.collate import TensorizedGraphData, collate_graphs

    g0 = TensorizedGraphData(np.zeros((3, 2), np.int32), np.ones(3, np.int32), [np.zeros((0, 2), np.int32)] * 2, {"candidate_nodes": np.array([1], np.int32)})
    g1 = TensorizedGraphData(np.zeros((1, 4), np.int32), np.ones(1, np.int32), [np.array([[0, 0]], np.int32), np.zeros((0, 2), np.int32)], {"candidate_nodes": np.zeros(0, np.int32)})
    gd = collate_graphs([g0, g1], 2)
    assert gd["token_ids"].shape == (4, 4) and gd["msg_src"].tolist() == [3] and gd["type_ptr"].tolist() == [0, 1, 1]
    assert gd["tgt_ptr"].tolist() == [0, 0, 0, 0, 1] and gd["candidate_ptr"].tolist() == [0, 1, 1]


def _model(hidden=32, **kw):
    from buglab.models.modelregistry import load_model

    return load_model({"modelName": "gnn-mlp", "hidden_state_size": hidden, "num_layers": 4, **kw}, Path("/tmp/_bl_test.pkl.gz"))[0]


def test_registry_contract():
    from buglab.models import modelregistry as R

    names = set(R.construct_model_dict(R.gnn, R.seq_transformer))
    assert names == {"gnn-mlp", "ggnn", "seq-great", "seq-rat", "seq-transformer", "seq-gru"}  # reference :129-137
    with pytest.raises(ValueError):
        R.load_model({"modelName": "nope"}, Path("/tmp/x.pkl.gz"))
    with pytest.raises(AssertionError):
        R.load_model({"modelName": "gnn-mlp"}, Path("/tmp/x.pt"))  # reference :147
    assert R.buggy_sample_weight_schedule(0.5)(3) == 0.5
    assert R.buggy_sample_weight_schedule("warmdown(10, 0.2)")(5) == pytest.approx(0.6)
    model, nn, init = R.load_model({"modelName": "gnn-mlp"}, Path("/tmp/x.pkl.gz"))
    assert nn is None and init is True
    assert model.gnn_model.node_representation_model.vocabulary_size == 15000 and model.gnn_model.max_nodes_per_graph == 35000


def test_node_embedder_dropout_comes_from_node_representations_only():
    """reference modelregistry.py:79-82: gnn() builds the node embedder as StrElementRepresentationModel(embedding_size=...,
    **node_representations) -- `dropout_rate` reaches the message-passing layers only.  The sequence models do hand their rate
    to the embedder (seqmodel.py:425-431)."""
    from buglab.models import modelregistry as R

    m, _, _ = R.load_model({"modelName": "gnn-mlp", "dropout_rate": 0.3}, Path("/tmp/x.pkl.gz"))
    assert m.gnn_model.node_representation_model.dropout_rate == 0.0
    m, _, _ = R.load_model({"modelName": "gnn-mlp", "dropout_rate": 0.3, "node_representations": {"dropout_rate": 0.15}}, Path("/tmp/x.pkl.gz"))
    assert m.gnn_model.node_representation_model.dropout_rate == 0.15
    m, _, _ = R.load_model({"modelName": "ggnn", "dropout_rate": 0.3}, Path("/tmp/x.pkl.gz"))
    assert m.gnn_model.node_representation_model.dropout_rate == 0.0


def test_message_activation_placement_reaches_every_layer_and_old_pickles_keep_theirs():
    """`message_activation_placement` enters through the registry's kwargs dict like `num_layers` (default "aggregated" =
    ptgnn's order as recollected); a module pickled before the switch existed has no such attribute
    and keeps the per-message placement it was trained with."""
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module
    from buglab.models.layers.messagepassing import MlpMessagePassingLayer

    assert hip_ops.message_activation_code("gelu") == "gelu_aggregated"
    assert hip_ops.message_activation_code("gelu", "message") == "gelu"
    assert hip_ops.message_activation_code("none", "aggregated") == "none"
    with pytest.raises(ValueError):
        hip_ops.message_activation_code("gelu", "sideways")
    with pytest.raises(ValueError):
        MlpMessagePassingLayer(8, 8, 8, 2, message_activation="swish")
    for kw, want in (({}, "gelu_aggregated"), ({"message_activation_placement": "message"}, "gelu"),
                     ({"message_activation": "none"}, "none")):
        creator = _model(**kw).gnn_model.message_passing_layer_creator
        layers = [l for l in creator(5) if isinstance(l, MlpMessagePassingLayer)]
        assert len(layers) == 4 and all(l._msg_act() == want for l in layers), (kw, [l._msg_act() for l in layers])
    m = build_gnn_mlp_module(32, 4, 3, 50, message_activation_placement="message")
    assert all(l._msg_act() == "gelu" for l in m._gnn.mp)
    old = build_gnn_mlp_module(32, 4, 3, 50)
    for l in old._gnn.mp:
        del l.__dict__["message_activation_placement"]  # what unpickling a round-5 checkpoint yields
        assert l._msg_act() == "gelu"


def test_message_width_limit_is_reported_at_construction():
    """hidden_state_size > 256 (message rows wider than 512 channels in the concat layers) is not built: said loudly when the
    layers are constructed, not by a failing kernel call in the first training step."""
    from buglab.models.gnn import build_gnn_mlp_module
    from buglab.models.layers.messagepassing import MlpMessagePassingLayer

    MlpMessagePassingLayer(512, 512, 256, 2)  # hidden 256: the widest shipped layer
    with pytest.raises(NotImplementedError, match="512"):
        MlpMessagePassingLayer(1024, 1024, 512, 2)
    with pytest.raises(NotImplementedError, match="hidden_state_size <= 256"):
        build_gnn_mlp_module(384, 4, 2, 50)


def test_tensorize_rewrite_bookkeeping_and_drop_rule():
    data = make_buglab_dataset(10, seed=4)
    model = _model()
    model.compute_metadata(copy.deepcopy(data))
    # 4 forward kinds + HasSubtoken, reversed, + self loops (gnn-mlp: add_self_edge=True)
    assert model.gnn_model.edge_types == ["Child", "HasSubtoken", "NextToken", "OccurrenceOf", "Sibling"]
    assert model.gnn_model.num_presented_edge_types == 11
    d = copy.deepcopy(data[0])
    t = model.tensorize(d)
    target = d["target_fix_action_idx"]
    scout = d["candidate_rewrite_metadata"][target][0]
    n_correct = sum(x is not None for x in (t.correct_rewrite_target, t.correct_candidate_symbol_node, t.correct_swapped_pair))
    assert n_correct == 1
    # training keeps only rewrites at the target location (reference basemodel.py:119-121)
    tnode = d["graph"]["reference_nodes"][target]
    kept = sorted(list(t.text_rewrite_original_idx) + list(t.candidate_rewrite_original_idx) + list(t.pair_rewrite_original_idx))
    assert kept == [i for i, n in enumerate(d["graph"]["reference_nodes"]) if n == tnode]
    if scout == "ArgSwapRewriteScout":
        assert t.correct_swapped_pair is not None
    with model._tensorize_all_location_rewrites():
        t_all = model.tensorize(copy.deepcopy(data[0]))
    assert len(t_all.text_rewrite_original_idx) + len(t_all.candidate_rewrite_original_idx) + len(t_all.pair_rewrite_original_idx) == len(d["candidate_rewrites"])
    model.gnn_model.max_nodes_per_graph = 10
    assert model.tensorize(copy.deepcopy(data[1])) is None  # reference gnn.py:404-405
    # self-loop type is last and has one edge per node
    model.gnn_model.max_nodes_per_graph = 35000
    assert t.graph_data.adjacency_lists[-1].shape[0] == t.graph_data.num_nodes


def test_minibatch_iterator_save_restore_and_unbatching(tmp_path):
    data = make_buglab_dataset(9, seed=5)
    model = _model()
    model.compute_metadata(copy.deepcopy(data))
    nn = model.build_neural_module()
    assert nn._gnn.mp[0].W.shape == (11, 64, 32) and nn._gnn.mp[3].W.shape == (11, 128, 64)
    mbs = list(model.minibatch_iterator(model.tensorize_dataset(copy.deepcopy(data), return_input_data=True), "cpu", max_minibatch_size=4))
    assert [len(o) for _, o in mbs] == [4, 4, 1]
    path = tmp_path / "m.pkl.gz"
    model.save(path, nn)
    m2, nn2 = type(model).restore_model(path, "cpu")
    assert all(torch.equal(a, b) for a, b in zip(nn.state_dict().values(), nn2.state_dict().values()))
    assert m2.gnn_model.edge_types == model.gnn_model.edge_types
    # un-batching (reference basemodel.py:240-346) with fabricated log-probabilities
    with model._tensorize_all_location_rewrites():
        mb, originals = next(iter(model.minibatch_iterator(model.tensorize_dataset(copy.deepcopy(data[:3]), return_input_data=True), "cpu", 50)))
    C_ = mb["graph_data"]["reference_node_ids"]["candidate_nodes"].shape[0]
    ids = torch.cat([mb["graph_data"]["reference_node_graph_idx"]["candidate_nodes"], torch.arange(3, dtype=torch.int32)])
    loc = torch.arange(C_ + 3, dtype=torch.float32) * -0.01
    text, var, swap = (torch.arange(mb[k].shape[0], dtype=torch.float32) + o for k, o in (("rewrite_to_location_group", 100.0), ("candidate_symbol_to_location_group", 200.0), ("swapped_pair_to_call_location_group", 300.0)))
    res = list(model._iter_per_sample_results(mb, ids, loc, swap, 3, originals, text, var))
    assert len(res) == 3
    for point, loc_lp, rewrite_lp in res:
        assert len(rewrite_lp) == len(point["candidate_rewrites"]) and -1 in loc_lp
        assert set(k for k in loc_lp if k >= 0) == set(point["graph"]["reference_nodes"])
        for (scout, _), lp in zip(point["candidate_rewrite_metadata"], rewrite_lp):
            assert (300 <= lp < 400) if scout == "ArgSwapRewriteScout" else ((200 <= lp < 300) if scout == "VariableMisuseRewriteScout" else (100 <= lp < 200))


def test_msgpack_roundtrip_and_split_identifier(tmp_path):
    from buglab.runtime.vocabulary import Vocabulary, split_identifier_into_parts
    from buglab.utils.msgpackutils import load_all_msgpack_l_gz, save_msgpack_l_gz

    data = make_buglab_dataset(3, seed=6)
    save_msgpack_l_gz(data, tmp_path / "a.msgpack.l.gz")
    back = list(load_all_msgpack_l_gz(str(tmp_path)))
    # (the native reader appends the subtoken nodes of data.py:97-121 right away; the Python reader's graph gets
    # them when as_graph_data() mutates it, like the reference's -- the file's own nodes come first either way)
    n0 = len(data[0]["graph"]["nodes"])
    assert len(back) == 3 and list(back[0]["graph"]["nodes"])[:n0] == data[0]["graph"]["nodes"]
    assert split_identifier_into_parts("getHTTPResponse_code2") == ["get", "http", "response", "code", "2"]
    assert split_identifier_into_parts("__") == ["__"]
    v = Vocabulary.create_vocabulary(["a", "a", "b"], max_size=10, count_threshold=2, add_unk=True)
    assert v.get_id_or_unk("a") != v.get_id_or_unk("zzz") == v.get_id_or_unk("b")


def test_hot_path_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a ROCm device the module must raise, not compute."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module

    m = build_gnn_mlp_module(32, 4, 3, 50)
    mb = C.to_device(C.collate_samples(make_samples(2, num_nodes=20, num_messages=50, num_edge_types=3, vocab_size=50, num_candidates=4), 3), "cpu")
    with pytest.raises(hip_ops.HipOpsUnavailable):
        m(**mb)


def test_ggnn_registry_recipe_shapes():
    """`ggnn` (reference modelregistry.py:132, gnnlayerdefs.py:42-68): 7 applications of ONE shared
    gated layer, concat residual, one gated layer on 2H-wide states; no self-loop edge type."""
    from buglab.models.layers.messagepassing import GatedMessagePassingLayer
    from buglab.models.modelregistry import load_model

    data = make_buglab_dataset(6, seed=7)
    model = load_model({"modelName": "ggnn", "hidden_state_size": 32}, Path("/tmp/_bl_ggnn.pkl.gz"))[0]
    model.compute_metadata(copy.deepcopy(data))
    assert model.gnn_model.num_presented_edge_types == 10  # 5 kinds + reversed, add_self_edge=False
    nn = model.build_neural_module()
    gated = [l for l in nn._gnn._recipe if isinstance(l, GatedMessagePassingLayer)]
    assert len(gated) == 8 and len({id(l) for l in gated[:7]}) == 1 and len(nn._gnn.mp) == 2
    assert nn._gnn.mp[0].W.shape == (10, 32, 32) and nn._gnn.mp[1].W.shape == (10, 64, 32) and nn._gnn.mp[1].Wh.shape == (64, 192)
    assert nn._gnn.output_node_state_dim == 64 and nn._localization_module.Ws.shape == (64, 64)


def test_token_occurrence_chunks():
    from buglab.data.collate import token_occurrence_chunks

    rng = np.random.default_rng(0)
    N, S = 50, 4
    ids = rng.integers(0, 9, (N, S)).astype(np.int32)
    lens = rng.integers(0, S + 1, N).astype(np.int32)
    occ, cptr, ctok = token_occurrence_chunks(ids, lens, chunk=5)
    assert occ.size == int(lens.sum()) and cptr[0] == 0 and cptr[-1] == occ.size and (np.diff(cptr) > 0).all() and (np.diff(cptr) <= 5).all()
    seen = set()
    for c in range(len(ctok)):
        for pos in occ[cptr[c]:cptr[c + 1]]:
            n, s = divmod(int(pos), S)
            assert s < lens[n] and ids[n, s] == ctok[c]
            seen.add(int(pos))
    assert len(seen) == occ.size  # every valid slot exactly once
    e = token_occurrence_chunks(np.zeros((3, 2), np.int32), np.zeros(3, np.int32))
    assert e[0].size == 0 and e[1].tolist() == [0] and e[2].size == 0


def test_node_order_puts_hubs_first():
    samples = make_samples(2, seed=3, num_nodes=300, num_messages=2000, num_edge_types=4, degree="powerlaw", max_degree=512)
    gd = C.collate_samples(samples, 4)["graph_data"]
    order = gd["node_order"]
    N = gd["token_ids"].shape[0]
    assert sorted(order.tolist()) == list(range(N))
    deg = np.diff(gd["tgt_ptr"]) + np.diff(gd["src_ptr"])
    k = int((deg > C.HUB_DEGREE).sum())
    assert k > 0 and (deg[order[:k]] > C.HUB_DEGREE).all() and (np.diff(deg[order[:k]]) <= 0).all()
    assert (np.diff(order[k:]) > 0).all()  # everyone else keeps the natural order


def test_parallel_minibatch_iterator_preserves_order(tmp_path):
    """Collation runs in several worker threads (runtime/neuralmodel.py): same minibatches, same order."""
    from buglab.models.modelregistry import load_model

    data = make_buglab_dataset(37, seed=2)
    model, _, _ = load_model({"modelName": "gnn-mlp"}, tmp_path / "m.pkl.gz")
    for d in copy.deepcopy(data):
        model.update_metadata_from(d)
    model.finalize_metadata()
    tensors = [(model.tensorize(d), i) for i, d in enumerate(copy.deepcopy(data))]
    seq = list(model.minibatch_iterator(iter(tensors), "cpu", 5, parallelize=False))
    par = list(model.minibatch_iterator(iter(tensors), "cpu", 5, parallelize=True))
    assert len(seq) == len(par) == 8 and [o for _, o in seq] == [o for _, o in par] == [list(range(i, min(i + 5, 37))) for i in range(0, 37, 5)]
    for (a, _), (b, _) in zip(seq, par):
        assert torch.equal(a["graph_data"]["msg_src"], b["graph_data"]["msg_src"]) and torch.equal(a["graph_data"]["token_ids"], b["graph_data"]["token_ids"])


def test_shard_loader_processes_yield_every_sample_once(tmp_path):
    """runtime/shardloader.py: worker processes read + tensorise whole shard files; together (and over the ranks of a
    data-parallel job) they yield exactly the samples of the in-process loader."""
    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset, tensorize_shards_parallel
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(45, seed=4)
    for i in range(5):
        save_msgpack_l_gz(data[9 * i : 9 * i + 9], tmp_path / f"s{i}.msgpack.l.gz")
    ds = ShardDataset(str(tmp_path))
    assert len(ds.shard_files()) == 5 and len(list(ds)) == 45
    model, _, _ = load_model({"modelName": "gnn-mlp"}, tmp_path / "m.pkl.gz")
    for d in ds:
        model.update_metadata_from(d)
    model.finalize_metadata()

    def key(t):
        return (t.graph_data.token_ids.tobytes(), tuple(a.tobytes() for a in t.graph_data.adjacency_lists))

    ref = sorted(key(model.tensorize(d)) for d in ds)
    got = sorted(key(t) for t in tensorize_shards_parallel(model, ds.shard_files(), num_workers=3))
    assert got == ref
    shares = [list(tensorize_shards_parallel(model, ds.shard_files(), num_workers=2, rank=r, world=2)) for r in range(2)]
    assert sorted(key(t) for s in shares for t in s) == ref and abs(len(shares[0]) - len(shares[1])) <= 5
    assert len(list(tensorize_shards_parallel(model, ds.shard_files(), num_workers=2, limit_num_yielded_elements=7))) == 7


def test_shard_loader_collated_minibatches(tmp_path):
    """Worker processes can also collate: whole NumPy minibatches cross the process boundary, every sample exactly once."""
    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset, collated_minibatches_parallel
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(40, seed=9)
    for i in range(4):
        save_msgpack_l_gz(data[10 * i : 10 * i + 10], tmp_path / f"s{i}.msgpack.l.gz")
    ds = ShardDataset(str(tmp_path))
    model, _, _ = load_model({"modelName": "gnn-mlp"}, tmp_path / "m.pkl.gz")
    for d in ds:
        model.update_metadata_from(d)
    model.finalize_metadata()
    mbs = list(collated_minibatches_parallel(model, ds.shard_files(), num_workers=2, max_minibatch_size=4))
    assert sum(int(m["graph_data"]["num_graphs"]) for m in mbs) == 40 and all(int(m["graph_data"]["num_graphs"]) <= 4 for m in mbs)
    ref_nodes = sorted(model.tensorize(d).graph_data.num_nodes for d in ds)
    assert sorted(int(n) for m in mbs for n in m["graph_data"]["num_nodes_per_graph"]) == ref_nodes
    dev = C.to_device(mbs[0], "cpu")
    assert dev["graph_data"]["msg_src"].dtype == torch.int32


def test_shard_loader_packed_minibatches_through_shared_memory(tmp_path):
    """packed=True: the int32 blob of every minibatch travels through a shared-memory segment; the result equals
    to_device() of the NumPy minibatch, and segments dropped by an early stop are unlinked."""
    import glob

    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset, collated_minibatches_parallel, receive_packed
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(24, seed=10)
    for i in range(3):
        save_msgpack_l_gz(data[8 * i : 8 * i + 8], tmp_path / f"s{i}.msgpack.l.gz")
    ds = ShardDataset(str(tmp_path))
    model, _, _ = load_model({"modelName": "gnn-mlp"}, tmp_path / "m.pkl.gz")
    for d in ds:
        model.update_metadata_from(d)
    model.finalize_metadata()
    before = set(glob.glob("/dev/shm/psm_*"))
    plain = {int(m["graph_data"]["msg_src"].sum()): C.to_device(m, "cpu") for m in collated_minibatches_parallel(model, ds.shard_files(), 1, 4)}
    got = [receive_packed(it, "cpu") for it in collated_minibatches_parallel(model, ds.shard_files(), 1, 4, packed=True)]
    assert len(got) == len(plain) == 6
    for g in got:
        ref = plain[int(g["graph_data"]["msg_src"].sum())]
        for k in ("msg_src", "msg_tgt", "type_ptr", "tgt_ptr", "tgt_msgs", "token_ids", "node_order", "head_gather_idx"):
            assert torch.equal(g["graph_data"][k], ref["graph_data"][k]), k
        assert torch.equal(g["has_bug"], ref["has_bug"]) and g["num_repair_groups"] == ref["num_repair_groups"]
        assert g["graph_data"]["head_spans"] == ref["graph_data"]["head_spans"]
    it = collated_minibatches_parallel(model, ds.shard_files(), 2, 2, packed=True)
    receive_packed(next(it), "cpu")
    it.close()  # early stop: pending segments are unlinked
    import time

    for _ in range(40):  # the workers' last puts drain asynchronously
        if set(glob.glob("/dev/shm/psm_*")) <= before:
            break
        time.sleep(0.05)
    assert set(glob.glob("/dev/shm/psm_*")) <= before


def test_prefetch_thread_stops_and_closes_its_source():
    import threading
    import time

    from buglab.runtime.trainer import _prefetch

    closed = threading.Event()

    def source():
        try:
            i = 0
            while True:
                yield i
                i += 1
        finally:
            closed.set()

    it = _prefetch(source(), depth=2)
    assert [next(it) for _ in range(5)] == [0, 1, 2, 3, 4]
    it.close()  # early stop (a rank whose peers ran out of minibatches)
    assert closed.wait(2.0)
    assert list(_prefetch(iter(range(7)), depth=3)) == list(range(7))

    def failing():
        yield 1
        raise ValueError("boom")

    it = _prefetch(failing())
    assert next(it) == 1
    with pytest.raises(ValueError):
        next(it)


def test_trainer_epoch_loop_with_loader_processes_on_cpu(tmp_path, monkeypatch):
    """ModelTrainer._run_training / _run_validation control flow with the multi-process loader, driven by a stand-in
    module on CPU tensors (the real module needs the GPU): every graph is seen once, loaders are shut down."""
    import glob

    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset
    from buglab.runtime.trainer import ModelTrainer
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(30, seed=12)
    for i in range(3):
        save_msgpack_l_gz(data[10 * i : 10 * i + 10], tmp_path / f"s{i}.msgpack.l.gz")
    ds = ShardDataset(str(tmp_path), shuffle=True)
    model, _, _ = load_model({"modelName": "gnn-mlp"}, tmp_path / "m.pkl.gz")
    for d in ds:
        model.update_metadata_from(d)
    model.finalize_metadata()

    class FakeModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(()))
            self.graphs = 0

        def forward(self, *, graph_data, has_bug, **_):
            assert graph_data["msg_src"].dtype == torch.int32 and graph_data["num_graphs"] == has_bug.shape[0]
            self.graphs += int(has_bug.shape[0])
            return self.w * 0.0 + float(graph_data["num_nodes"])

        def reset_metrics(self):
            pass

        def report_metrics(self):
            return {"Loss": 0.0}

    class FakeOpt:
        def __init__(self):
            self.steps = 0

        def zero_grad(self):
            pass

        def step(self, weight=1.0):
            self.steps += 1

    monkeypatch.setenv("BUGLAB_LOADER_WORKERS", "2")
    trainer = ModelTrainer(model, tmp_path / "m.pkl.gz", minibatch_size=4)
    trainer.neural_module = FakeModule()
    trainer._use_multiprocessing = True
    before = set(glob.glob("/dev/shm/psm_*"))
    opt = FakeOpt()
    trainer._run_training(ds, 0, torch.device("cpu"), opt, None, True)
    assert trainer.neural_module.graphs == 30 and opt.steps >= 8
    trainer.neural_module.graphs = 0
    trainer._run_validation(ds, 0, float("inf"), torch.device("cpu"), True, False)
    assert trainer.neural_module.graphs == 30
    assert set(glob.glob("/dev/shm/psm_*")) <= before
    # the next epoch's loaders forked ahead of time (what train() does before validation): the epoch picks the pool up, sees
    # every graph once, and leaves nothing behind; a pool nobody consumes is shut down by _drop_prestarted_pool
    trainer._prestart_loaders(ds, 1, True)
    pool = trainer._prestarted[2]
    assert pool._procs is not None and all(p.is_alive() or p.exitcode == 0 for p in pool._procs)
    trainer.neural_module.graphs = 0
    trainer._run_training(ds, 1, torch.device("cpu"), FakeOpt(), None, True)
    assert trainer.neural_module.graphs == 30 and trainer._prestarted is None and pool._closed
    # a pool that ran to its end is reaped by a background thread, not on the trainer's thread: the workers are gone shortly after
    assert pool._finished
    import time as _time

    deadline = _time.monotonic() + 10.0
    while any(p.is_alive() for p in pool._procs) and _time.monotonic() < deadline:
        _time.sleep(0.05)
    assert not any(p.is_alive() for p in pool._procs)
    # the order train() really runs: prestart (next epoch) -> validation -> training.  Validation -- over another dataset
    # object or over the very same one -- must neither consume nor close the pool; the training epoch then reuses it.
    ds_val = ShardDataset(str(tmp_path), shuffle=False)
    for val in (ds_val, ds):
        trainer._prestart_loaders(ds, 3, True)
        pool = trainer._prestarted[2]
        trainer.neural_module.graphs = 0
        trainer._run_validation(val, 2, float("inf"), torch.device("cpu"), True, False)
        assert trainer.neural_module.graphs == 30
        assert trainer._prestarted is not None and trainer._prestarted[2] is pool and not pool._closed
        trainer.neural_module.graphs = 0
        trainer._run_training(ds, 3, torch.device("cpu"), FakeOpt(), None, True)
        assert trainer.neural_module.graphs == 30 and trainer._prestarted is None and pool._finished
    trainer._prestart_loaders(ds, 2, True)
    pool = trainer._prestarted[2]
    trainer._drop_prestarted_pool()
    assert pool._closed and not any(p.is_alive() for p in pool._procs)
    assert set(glob.glob("/dev/shm/psm_*")) <= before
    # the in-process path (loader workers off) sees the same graphs
    monkeypatch.setenv("BUGLAB_LOADER_WORKERS", "0")
    trainer.neural_module.graphs = 0
    trainer._run_training(ds, 1, torch.device("cpu"), FakeOpt(), None, True)
    assert trainer.neural_module.graphs == 30


def test_optimizer_state_travels_in_a_sidecar_next_to_the_checkpoint(tmp_path):
    """Adam moments + step count (warm-up position) are written to `<checkpoint>.optim` with the best checkpoint and
    picked up when training continues from it; a sidecar of another architecture is ignored, a missing one is fine."""
    import torch

    from buglab.runtime.optim import FlatAdam
    from buglab.runtime.trainer import ModelTrainer

    def make(n=7):
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(n, 3)), torch.nn.Parameter(torch.randn(5))]
        return FlatAdam(ps, lr=1e-3, num_warmup_steps=10, distributed=False)

    ckpt = tmp_path / "model.pkl.gz"
    trainer = ModelTrainer(model=None, save_location=ckpt)
    opt = make()
    opt.m.uniform_(-1, 1)
    opt.v.uniform_(0, 1)
    opt.step_count = 42
    trainer._save_optimizer_state(opt)
    assert (tmp_path / "model.pkl.gz.optim").exists() and not (tmp_path / "model.pkl.gz.optim.tmp").exists()

    fresh = make()
    other = ModelTrainer(model=None, save_location=tmp_path / "elsewhere.pkl.gz")
    other._restore_optimizer_state(fresh, "cpu")  # nothing asked for: untouched
    assert fresh.step_count == 0
    other.restore_optimizer_state_from = ckpt
    other._restore_optimizer_state(fresh, "cpu")
    assert fresh.step_count == 42 and torch.equal(fresh.m, opt.m) and torch.equal(fresh.v, opt.v)

    mismatched = make(n=9)
    other._restore_optimizer_state(mismatched, "cpu")  # different parameter count: ignored with a warning
    assert mismatched.step_count == 0 and float(mismatched.m.abs().sum()) == 0.0
    other.restore_optimizer_state_from = tmp_path / "missing.pkl.gz"
    other._restore_optimizer_state(fresh, "cpu")
    assert fresh.step_count == 42


def test_deterministic_mode_keeps_every_token_in_one_chunk(monkeypatch):
    """BL_DETERMINISTIC=1: the embedding-gradient kernel adds once per table row (one chunk per token) -- both collators."""
    from buglab.data import collate as C
    from buglab.data.synthetic import make_samples

    samples = make_samples(6, seed=3, num_nodes=400, num_messages=1500, num_edge_types=4, vocab_size=20)  # few tokens: hot rows
    monkeypatch.delenv("BL_DETERMINISTIC", raising=False)
    free = C.collate_samples(samples, 4)["graph_data"]
    assert len(free["tok_chunk_id"]) > len(np.unique(free["tok_chunk_id"]))  # hot tokens are cut into chunks of <= 256
    monkeypatch.setenv("BL_DETERMINISTIC", "1")
    det = C.collate_samples(samples, 4)["graph_data"]
    assert len(det["tok_chunk_id"]) == len(np.unique(det["tok_chunk_id"]))
    assert np.array_equal(np.sort(det["tok_occ"]), np.sort(free["tok_occ"]))  # the same occurrences, differently grouped
    assert det["tok_chunk_ptr"][-1] == len(det["tok_occ"])

"""Edge features (`edge_feature_size` > 0, reference modelregistry.py:56,70-86, gnnlayerdefs.py:13,22, data.py:158-161):
host side on the CPU (vocabulary, per-message feature ids through tensorize / collate / packing), the oracle's
gradient by finite differences, and -- on the GPU -- the HIP path against the oracle.  ptgnn's own treatment is not in
the reference tree: the spec is the one frozen in DESIGN.md section 2 (reversed edges carry their forward edge's
feature, self loops the pad token)."""
import numpy as np
import pytest
import torch

from oracle import buglab_oracle as O


def _graph(rng, n, kinds, with_feats=True):
    from buglab.representations.data import GraphData
    from buglab.runtime.vocabulary import Vocabulary

    edges, feats = {}, {}
    for k in kinds:
        m = int(rng.integers(3, 12))
        e = np.stack([rng.integers(0, n, m), rng.integers(0, n, m)], 1).astype(np.int32)
        edges[k] = np.unique(e, axis=0)  # unique (src, tgt) pairs: a message identifies its edge
        feats[k] = [str(rng.choice(["arg0", "arg1", "kw", Vocabulary.get_pad()])) for _ in range(edges[k].shape[0])]
    return GraphData(node_information=[f"name{i % 7}" for i in range(n)], edges=edges, reference_nodes={"candidate_nodes": np.arange(3, dtype=np.int32)},
                     edge_features=feats if with_feats else None)


def test_feature_ids_follow_their_edges_through_tensorize_collate_and_packing():
    from buglab.data.collate import collate_graphs, pack_minibatch
    from buglab.models.graphmodel import GraphNeuralNetworkModel, StrElementRepresentationModel
    from buglab.runtime.vocabulary import Vocabulary

    rng = np.random.default_rng(0)
    kinds = ["Child", "NextToken", "Arg"]
    graphs = [_graph(rng, 20 + 3 * i, kinds) for i in range(6)]
    model = GraphNeuralNetworkModel(
        node_representation_model=StrElementRepresentationModel(embedding_size=16, min_freq_threshold=1),
        edge_representation_model=StrElementRepresentationModel(token_splitting="token", embedding_size=8, min_freq_threshold=1),
        add_self_edges=True, message_passing_layer_creator=lambda n: [])
    for g in graphs:
        model.update_metadata_from(g)
    model.finalize_metadata()
    voc = model.edge_representation_model.vocabulary
    pad_id = voc.get_id_or_unk(Vocabulary.get_pad())
    assert {"arg0", "arg1", "kw"} <= set(voc.id_to_token) and voc.get_id_or_unk("never seen") == voc.get_id_or_unk(Vocabulary.get_unk())
    tens = [model.tensorize(g) for g in graphs]
    T = model.num_presented_edge_types
    assert T == 2 * len(kinds) + 1
    for g, t in zip(graphs, tens):
        assert len(t.edge_feature_ids) == T and [f.shape[0] for f in t.edge_feature_ids] == [a.shape[0] for a in t.adjacency_lists]
        for j, k in enumerate(model.edge_types):
            want = [voc.get_id_or_unk(s) for s in g.edge_features[k]]
            assert t.edge_feature_ids[j].tolist() == want                      # forward edges: their own token
            assert t.edge_feature_ids[len(kinds) + j].tolist() == want         # reversed edges: the forward edge's token
        assert (t.edge_feature_ids[-1] == pad_id).all()                        # self loops: the pad token
    gd = collate_graphs(tens, T)
    E = gd["msg_src"].shape[0]
    assert gd["msg_feat"].shape == (E,) and gd["msg_feat"].dtype == np.int32
    # every message finds its edge's feature id again (messages are type-major, target-sorted: not the edge-list order)
    off = np.concatenate([[0], np.cumsum([t.num_nodes for t in tens])])
    table = {}
    for b, t in enumerate(tens):
        for ty, (adj, f) in enumerate(zip(t.adjacency_lists, t.edge_feature_ids)):
            for (s, d), fid in zip(adj.tolist(), f.tolist()):
                table[(ty, s + off[b], d + off[b])] = fid
    tptr = gd["type_ptr"]
    for ty in range(T):
        for e in range(tptr[ty], tptr[ty + 1]):
            assert table[(ty, gd["msg_src"][e], gd["msg_tgt"][e])] == gd["msg_feat"][e]
    # the packed blob carries it (what the loader processes hand to the trainer)
    mb = {"graph_data": gd, "has_bug": np.zeros(len(tens), np.int32), "num_repair_groups": 0,
          "text_rewrite_original_idxs": [], "candidate_rewrite_original_idxs": [], "pair_rewrite_original_idx": []}
    from buglab.data import collate as C

    for k in C._INT_KEYS_MB:
        mb[k] = np.zeros(0, np.int32)
    blob, meta = pack_minibatch(mb)
    where = {k: (shape, o) for w, k, shape, o in meta["layout"] if w == "gd"}
    shape, o = where["msg_feat"]
    assert np.array_equal(blob[o : o + E], gd["msg_feat"])
    # a model without the edge model produces no feature ids, and the collator no msg_feat
    plain = GraphNeuralNetworkModel(node_representation_model=StrElementRepresentationModel(embedding_size=16, min_freq_threshold=1),
                                    add_self_edges=True, message_passing_layer_creator=lambda n: [])
    for g in graphs:
        plain.update_metadata_from(g)
    plain.finalize_metadata()
    assert plain.tensorize(graphs[0]).edge_feature_ids is None and "msg_feat" not in collate_graphs([plain.tensorize(g) for g in graphs], T)


def _edge_feature_golden():
    import json
    import os

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(gdir, "edge_features.json"), encoding="utf-8") as f:
        return os.path.join(gdir, "reference_shard_edge_features.msgpack.l.gz"), json.load(f)["datapoints"]


@pytest.mark.parametrize("reader", ["python", "native"])
def test_edge_features_of_as_graph_data_match_the_reference(reader, monkeypatch):
    """`GraphData.edge_features` against what the REFERENCE's own `as_graph_data` produced for the same shard
    (tests/golden/make_golden_edge_features.py; reference data.py:158-161): labels, pads, empty and non-ASCII labels, edge order
    -- on the Python reader path and on the native one (where the strings stay in the reader's blob until asked for)."""
    from buglab.data import native
    from buglab.representations.data import BugLabData
    from buglab.utils.msgpackutils import load_msgpack_l_gz

    shard, want = _edge_feature_golden()
    if reader == "native":
        if not native.available():
            pytest.skip("libbuglab_data.so not built")
        points = list(native.load_msgpack_l_gz_native(shard))
        assert isinstance(points[0]["graph"], native.NativeGraph)
    else:
        monkeypatch.setenv("BUGLAB_NATIVE_READER", "0")
        points = list(load_msgpack_l_gz(shard))
        assert not isinstance(points[0]["graph"], native.NativeGraph)
    assert len(points) == len(want) == 8
    for d, w in zip(points, want):
        gd, _ = BugLabData.as_graph_data(d)
        assert set(gd.edge_features.keys()) == set(w.keys()) == set(gd.edges.keys())
        for kind in w:
            assert list(gd.edge_features[kind]) == w[kind], kind
            assert len(w[kind]) == np.asarray(gd.edges[kind]).reshape(-1, 2).shape[0]


def _case_with_features(F=8, V=11, **kw):
    from buglab.data.collate import collate_samples
    from tests import helpers as Hh

    cfg, samples, _ = Hh.make_case(**kw)
    cfg.edge_feature_size, cfg.edge_vocab_size = F, V
    rng = np.random.default_rng(5)
    for s in samples:
        s.graph_data.edge_feature_ids = [rng.integers(0, V, a.shape[0]).astype(np.int32) for a in s.graph_data.adjacency_lists]
    return cfg, samples, collate_samples(samples, cfg.num_edge_types)


def test_oracle_edge_feature_gradient_by_finite_differences():
    cfg, _, mb = _case_with_features(B=2, n=30, E=120, T=3, H=16, layers=4, vocab=60, C=4, seed=2)
    params = {k: v.double() for k, v in O.init_params(cfg, seed=0).items()}
    assert params["mp.0.W"].shape == (3, 2 * 16 + 8, 16) and params["mp.3.W"].shape == (3, 2 * 32 + 8, 32)
    out, grads = O.forward_backward(params, mb, cfg)
    used = np.unique(mb["graph_data"]["msg_feat"])
    g = grads["edge_embed.table"]
    assert float(g.abs().sum()) > 0 and float(g[[i for i in range(11) if i not in used]].abs().sum()) == 0.0
    i, j = int(used[0]), 3
    eps = 1e-6
    vals = []
    for sgn in (+1, -1):
        p2 = {k: v.clone() for k, v in params.items()}
        p2["edge_embed.table"][i, j] += sgn * eps
        vals.append(float(O.forward_loss(p2, mb, cfg)["loss"]))
    num = (vals[0] - vals[1]) / (2 * eps)
    assert abs(num - float(g[i, j])) < 1e-6 + 1e-4 * abs(num), (num, float(g[i, j]))


def test_registry_builds_the_edge_feature_model():
    from buglab.models.modelregistry import gnn
    from buglab.models.gnnlayerdefs import create_ggnn_mp_layers, create_mlp_mp_layers

    model = gnn(mp_layer=create_mlp_mp_layers, add_self_edge=True, hidden_state_size=32, edge_feature_size=8, num_layers=4)
    em = model._gnn_model.edge_representation_model
    assert em is not None and em.token_splitting == "token" and em.embedding_size == 8
    with pytest.raises(NotImplementedError):
        gnn(mp_layer=create_ggnn_mp_layers, add_self_edge=False, hidden_state_size=32, edge_feature_size=8)


def test_host_pipeline_tags_the_args_edges_of_raw_datapoints():
    """Raw BugLabData (Child edges of Call nodes carry the third element "args", reference data.py:158-161) through the
    registry model's metadata pass, tensorize and minibatch building: the messages of those edges -- forward and reversed --
    carry the "args" token's id, every other message the pad token's."""
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models.gnnlayerdefs import create_mlp_mp_layers
    from buglab.models.modelregistry import gnn
    from buglab.runtime.vocabulary import Vocabulary

    data = make_buglab_dataset(12, seed=3)
    model = gnn(mp_layer=create_mlp_mp_layers, add_self_edge=True, hidden_state_size=32, edge_feature_size=8, num_layers=4)
    for d in data:
        model.update_metadata_from(d)
    model.finalize_metadata()
    gm = model._gnn_model
    voc = gm.edge_representation_model.vocabulary
    args_id, pad_id = voc.get_id_or_unk("args"), voc.get_id_or_unk(Vocabulary.get_pad())
    assert args_id not in (pad_id, voc.get_id_or_unk(Vocabulary.get_unk()))
    mb = model.initialize_minibatch()
    n_args = 0
    for d in data:
        t = model.tensorize(d)
        assert t is not None
        n_args += sum(1 for e in d["graph"]["edges"]["Child"] if len(e) >= 3)
        model.extend_minibatch_with(t, mb)
    out = model.finalize_minibatch(mb, "cpu")
    gd = out["graph_data"]
    feat = np.asarray(gd["msg_feat"].cpu() if hasattr(gd["msg_feat"], "cpu") else gd["msg_feat"])
    tptr = np.asarray(gd["type_ptr_host"])
    child = gm.edge_types.index("Child")
    nk = len(gm.edge_types)
    for ty in range(len(tptr) - 1):
        seg = feat[tptr[ty] : tptr[ty + 1]]
        if ty in (child, nk + child):
            assert int((seg == args_id).sum()) == n_args and set(np.unique(seg).tolist()) <= {args_id, pad_id}
        else:
            assert (seg == pad_id).all()


def test_loader_processes_hand_over_the_feature_ids(tmp_path):
    """Shards on disk -> loader processes (native reader, tensorise, native collator, packed through shared memory) ->
    `receive_packed`: `msg_feat` arrives with the other index arrays and equals what the in-process NumPy path builds."""
    from buglab.data import collate as C
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset, collated_minibatches_parallel, receive_packed
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(24, seed=12)
    for i in range(3):
        save_msgpack_l_gz(data[8 * i : 8 * i + 8], tmp_path / f"s{i}.msgpack.l.gz")
    ds = ShardDataset(str(tmp_path))
    model, _, _ = load_model({"modelName": "gnn-mlp", "edge_feature_size": 8, "hidden_state_size": 32, "num_layers": 4}, tmp_path / "m.pkl.gz")
    for d in ds:
        model.update_metadata_from(d)
    model.finalize_metadata()
    args_id = model._gnn_model.edge_representation_model.vocabulary.get_id_or_unk("args")
    plain = {int(m["graph_data"]["msg_src"].sum()): C.to_device(m, "cpu") for m in collated_minibatches_parallel(model, ds.shard_files(), 1, 4)}
    got = [receive_packed(it, "cpu") for it in collated_minibatches_parallel(model, ds.shard_files(), 2, 4, packed=True)]
    assert len(got) == len(plain) == 6
    n_args = 0
    for g in got:
        ref = plain[int(g["graph_data"]["msg_src"].sum())]
        for k in ("msg_src", "msg_tgt", "type_ptr", "msg_feat"):
            assert torch.equal(g["graph_data"][k], ref["graph_data"][k]), k
        assert g["graph_data"]["msg_feat"].dtype == torch.int32 and g["graph_data"]["msg_feat"].shape == g["graph_data"]["msg_src"].shape
        n_args += int((g["graph_data"]["msg_feat"] == args_id).sum())
    assert n_args == 2 * sum(1 for d in data for e in d["graph"]["edges"]["Child"] if len(e) >= 3)  # forward + reversed messages


@pytest.mark.gpu
def test_train_cli_with_edge_features(tmp_path):
    """`train.py gnn-mlp ... --model-spec '{"edge_feature_size": 8}'` end to end (loader processes, trainer, evaluation)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models import evaluate, train
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(64, seed=1)
    (tmp_path / "train").mkdir()
    (tmp_path / "valid").mkdir()
    save_msgpack_l_gz(data[:48], tmp_path / "train" / "a.msgpack.l.gz")
    save_msgpack_l_gz(data[48:], tmp_path / "valid" / "v.msgpack.l.gz")
    model_path = tmp_path / "model.pkl.gz"
    args = train.parse_args(["gnn-mlp", str(tmp_path / "train"), str(tmp_path / "valid"), str(model_path), "--max-num-epochs", "2",
                             "--minibatch-size", "16", "--quiet", "--model-spec",
                             '{"hidden_state_size": 64, "num_layers": 4, "edge_feature_size": 8}'])
    train.run(args)
    metrics = evaluate.run({"MODEL_FILENAME": str(model_path), "TEST_DATA_PATH": str(tmp_path / "valid"), "--assume-buggy": False,
                            "--eval-only-no-bug": False, "--limit-num-elements": None, "--sequential": True})
    assert metrics["num_samples"] == 16 and 0.0 <= metrics["localization_accuracy"] <= 1.0
    from buglab.models.gnn import GnnBugLabModel

    model, nn = GnnBugLabModel.restore_model(model_path, torch.device("cuda"))
    assert nn._gnn.edge_embed is not None and nn._gnn.mp[0].W.shape[1] == 2 * 64 + 8


@pytest.mark.gpu
@pytest.mark.parametrize("dropout,seed,F", [(0.0, None, 8), (0.2, 31, 8), (0.0, None, 32), (0.2, 31, 64)])
def test_edge_feature_model_matches_oracle_on_gpu(dropout, seed, F):
    """Loss, node states and every gradient (incl. the edge-embedding table and the [T, 2 Din + F, Dm] message weights).
    F = 8: the exact-fp32 GEMM kernels; F = 32 / 64 (a multiple of 32 like the node widths): the bf16x6 kernels with the packed
    table as the third gathered source of the forward, weight-gradient and routed input-gradient GEMMs."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from buglab.models import hip_ops
    from tests.test_hip_parity import _check_against_oracle, _check_tie_aware

    assert hip_ops.x6_ok(64, 64, F) == (F % 32 == 0)
    cfg, _, mb = _case_with_features(F=F, B=3, n=70, E=350, T=5, H=64, layers=4, vocab=200, C=6, seed=4, dropout=dropout)
    if F % 32 == 0:
        # split-precision products round differently from the fp32 oracle's: a near-tie of the max may route a channel to the
        # other message (an O(1) change of that channel's gradient) -- compared tie-aware against the fp64 oracle, like
        # every bf16x6 configuration in tests/test_hip_parity.py (winner tables equal up to true near-ties, gradients at 1e-4
        # with the routing injected; the table and the [T, 2 Din + F, Dm] weights are in the compared set)
        params = __import__("oracle.buglab_oracle", fromlist=["x"]).init_params(cfg, seed=0)
        assert "edge_embed.table" in params
        flips, total = _check_tie_aware(cfg, mb, seed=seed)
        assert flips <= 2e-3 * total
        return
    module, out, worst = _check_against_oracle(cfg, mb, seed=seed)
    assert "edge_embed.table" in worst and worst["edge_embed.table"][1] > 0

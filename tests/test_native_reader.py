"""CPU: the native (C++) `*.msgpack.l.gz` reader against the Python reader of the reference's format
(SURVEY.md section 8f rank 4): identical datapoints, identical open-vocabulary nodes / HasSubtoken edges,
identical tensorised samples, and the error behaviour of the reference's loader (bad files are skipped)."""
import gzip
import os
import re
import subprocess

import msgpack
import numpy as np
import pytest

from buglab.data import native
from buglab.data.synthetic import make_buglab_dataset
from buglab.representations.data import BugLabData, add_open_vocab_nodes_and_edges
from buglab.runtime.vocabulary import split_identifier_into_parts
from buglab.utils.msgpackutils import load_all_msgpack_l_gz, load_msgpack_l_gz, save_msgpack_l_gz
from tests.conftest import PKG, ROOT


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not native.available():
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc_data")], check=True)
    native.load_library()


def _weird_datapoint():
    d = make_buglab_dataset(1, seed=5)[0]
    g = d["graph"]
    n0 = len(g["nodes"])
    g["nodes"] += ["HTTPServer2_fooBar", "__init__", "naïveÉcole_x", "x", "A", "aB", "ABc", "mixed9Case99", "_", "", "snake_case_name", "ünïcode"]
    g["edges"]["NextToken"] += [[n0 + i, n0 + i + 1] for i in range(11)]
    g["edges"]["Empty"] = []
    g["edges"]["Child"].append([0, 1, "body"])
    d["candidate_rewrite_logprobs"] = [0.25] * len(d["candidate_rewrites"])
    d["extra"] = {"nested": [1, 2.5, None, True, b"\x00\x01", {"k": -7, "big": 2 ** 40}], "neg": -129, "f32": 1.5}
    return d


def test_exports_match_header():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "buglab_data.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(bl_[a-z0-9_]+)\s*\(", src)))
    nm = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert declared == sorted(set(re.findall(r" T (bl_[a-z0-9_]+)", nm))) == sorted(native.EXPORTED_SYMBOLS)
    assert native.load_library().bl_data_version() >= 3  # 3: per-edge payload in bl_collate_graphs


def test_subtoken_split_matches_python_regex():
    lib = native.load_library()
    rng = np.random.default_rng(0)
    alphabet = list("abcXYZ019_-.$") + ["Ab", "HTTP", "x"]
    words = ["".join(rng.choice(alphabet, size=int(rng.integers(0, 9)))) for _ in range(3000)] + ["", "_", "__", "A", "aB", "ABc", "ABCd9e", "a__b"]
    vocab_tokens = sorted({p for w in words for p in split_identifier_into_parts(w)})[::2]  # half the parts are unknown
    tok = ["%PAD%", "%UNK%"] + vocab_tokens
    nv = native.NativeVocabulary(tok, 1)
    enc = [w.encode() for w in words]
    off = np.zeros(len(enc) + 1, dtype=np.int32)
    np.cumsum([len(e) for e in enc], out=off[1:])
    ids, lens, needs = nv.tensorize(native.NativeNodes(b"".join(enc), off), 6)
    lookup = {t: i for i, t in enumerate(tok)}
    assert not needs.any()
    for i, w in enumerate(words):
        parts = split_identifier_into_parts(w)[:6]
        assert lens[i] == max(1, len(parts)), w
        assert ids[i, : len(parts)].tolist() == [lookup.get(p, 1) for p in parts], (w, parts)
        assert (ids[i, len(parts):] == 0).all()
    assert lib.bl_data_last_error() is not None


def test_native_reader_yields_the_same_datapoints(tmp_path):
    data = make_buglab_dataset(40, seed=3) + [None, _weird_datapoint()]
    path = str(tmp_path / "shard.msgpack.l.gz")
    save_msgpack_l_gz(data, path)
    py = list(load_msgpack_l_gz(path, native=False))
    nat = list(load_msgpack_l_gz(path, native=True))
    assert len(py) == len(nat) == 42 and nat[40] is None and py[40] is None
    for a, b in zip(py, nat):
        if a is None:
            continue
        ga = dict(a["graph"])
        if "HasSubtoken" not in ga["edges"]:
            add_open_vocab_nodes_and_edges(ga)  # what as_graph_data does on the Python path
        gb = b["graph"]
        assert list(gb["nodes"]) == ga["nodes"]
        assert list(gb["edges"].keys()) == list(ga["edges"].keys())
        for k in ga["edges"]:
            assert [list(e) for e in gb["edges"][k]] == [list(e) for e in ga["edges"][k]], k
        assert list(gb["reference_nodes"]) == list(ga["reference_nodes"])
        for k in ("path", "text", "code_range"):
            assert gb[k] == ga[k] or list(map(list, gb[k])) == list(map(list, ga[k]))
        for k in a:
            if k != "graph":
                assert msgpack.packb(a[k]) == msgpack.packb(b[k]), k
    # the unicode identifier forced the Python fallback for that one datapoint only
    assert isinstance(nat[0]["graph"], native.NativeGraph) and not isinstance(nat[41]["graph"], native.NativeGraph)


def test_tensorised_samples_are_identical(tmp_path):
    from buglab.models.modelregistry import load_model

    data = make_buglab_dataset(30, seed=8)
    path = str(tmp_path / "train.msgpack.l.gz")
    save_msgpack_l_gz(data, path)
    model, _, _ = load_model({"modelName": "gnn-mlp"}, tmp_path / "m.pkl.gz")
    for d in load_msgpack_l_gz(path, native=False):
        model.update_metadata_from(d)
    model.finalize_metadata()
    for a, b in zip(load_msgpack_l_gz(path, native=False), load_msgpack_l_gz(path, native=True)):
        ta, tb = model.tensorize(a), model.tensorize(b)
        ga, gb = ta.graph_data, tb.graph_data
        assert np.array_equal(ga.token_ids, gb.token_ids) and np.array_equal(ga.token_lens, gb.token_lens)
        assert len(ga.adjacency_lists) == len(gb.adjacency_lists)
        for x, y in zip(ga.adjacency_lists, gb.adjacency_lists):
            assert np.array_equal(x, y) and y.dtype == np.int32
        assert ga.reference_nodes.keys() == gb.reference_nodes.keys()
        for k in ga.reference_nodes:
            assert np.array_equal(np.asarray(ga.reference_nodes[k]), np.asarray(gb.reference_nodes[k])), k
        for f in ta._fields:
            if f != "graph_data":
                x, y = getattr(ta, f), getattr(tb, f)
                assert (x is None and y is None) or np.array_equal(np.asarray(x, dtype=object), np.asarray(y, dtype=object)), f


def test_bad_files_are_reported_and_skipped(tmp_path, capsys):
    good = make_buglab_dataset(3, seed=1)
    save_msgpack_l_gz(good, str(tmp_path / "a.msgpack.l.gz"))
    whole = b"".join(msgpack.packb(d, use_bin_type=True) for d in good)
    with gzip.open(str(tmp_path / "b.msgpack.l.gz"), "wb") as f:
        f.write(whole[: len(whole) - 17])  # truncated in the middle of the last object
    with open(str(tmp_path / "c.msgpack.l.gz"), "wb") as f:
        f.write(b"this is not gzip")
    got = list(load_all_msgpack_l_gz(str(tmp_path)))
    assert len(got) == 3 + 2  # the reference's loader prints the error and goes on (msgpackutils.py:45-46)
    assert "Error loading" in capsys.readouterr().out
    with pytest.raises((OSError, ValueError)):
        list(native.load_msgpack_l_gz_native(str(tmp_path / "missing.msgpack.l.gz")))


@pytest.mark.parametrize("degree", ["uniform", "powerlaw"])
def test_native_collator_equals_numpy_collator(degree, monkeypatch):
    """bl_collate_graphs (one GIL-free call) against the NumPy collator, array by array."""
    from buglab.data import collate as C
    from buglab.data.synthetic import make_samples

    samples = make_samples(7, seed=11, num_nodes=180, num_messages=1100, num_edge_types=5, degree=degree, max_degree=300)
    samples[3].graph_data.adjacency_lists[2] = np.zeros((0, 2), np.int32)   # an empty edge type in one graph
    samples[5].graph_data.token_lens[:7] = 0                                # nodes without subtokens
    monkeypatch.setenv("BUGLAB_NATIVE_COLLATE", "1")
    a = C.collate_samples(samples, 6)["graph_data"]                         # 6 presented types: the last one is empty everywhere
    monkeypatch.setenv("BUGLAB_NATIVE_COLLATE", "0")
    b = C.collate_samples(samples, 6)["graph_data"]
    for k in ("token_ids", "token_lens", "msg_src", "msg_tgt", "type_ptr", "tgt_ptr", "tgt_msgs", "src_ptr", "src_msgs", "node_order",
              "tok_occ", "tok_chunk_ptr", "tok_chunk_id"):
        assert a[k].dtype == np.int32 and np.array_equal(a[k], b[k]), k
    assert (np.diff(a["tgt_ptr"]) + np.diff(a["src_ptr"])).max() > C.HUB_DEGREE or degree == "uniform"
    assert "msg_feat" not in a and "msg_feat" not in b
    # a per-edge payload (edge-feature token ids, edge_feature_size > 0) follows its message through both collators
    rng = np.random.default_rng(3)
    for smp in samples:
        smp.graph_data.edge_feature_ids = [rng.integers(0, 1000, adj.shape[0]).astype(np.int32) for adj in smp.graph_data.adjacency_lists]
    monkeypatch.setenv("BUGLAB_NATIVE_COLLATE", "1")
    a = C.collate_samples(samples, 6)["graph_data"]
    monkeypatch.setenv("BUGLAB_NATIVE_COLLATE", "0")
    b = C.collate_samples(samples, 6)["graph_data"]
    assert a["msg_feat"].dtype == np.int32 and a["msg_feat"].shape == a["msg_src"].shape
    for k in ("msg_src", "msg_tgt", "type_ptr", "msg_feat"):
        assert np.array_equal(a[k], b[k]), k
    samples[2].graph_data.edge_feature_ids = None  # a graph without ids in a minibatch that has them: refused by both
    for flag in ("1", "0"):
        monkeypatch.setenv("BUGLAB_NATIVE_COLLATE", flag)
        with pytest.raises((ValueError, AssertionError)):
            C.collate_samples(samples, 6)
    for smp in samples:
        smp.graph_data.edge_feature_ids = None
    # an empty minibatch of graphs without edges
    empty = C.collate_graphs([], 4)
    assert empty["msg_src"].size == 0 and empty["type_ptr"].tolist() == [0] * 5


def test_shard_written_by_the_reference_reads_back_identically(tmp_path):
    """On-disk format, pinned: `tests/golden/reference_shard.msgpack.l.gz` was written by the reference's own
    `save_msgpack_l_gz` and `reference_shard.json` is what its `load_msgpack_l_gz` reads back
    (tests/golden/make_golden_shard.py).  Both readers here must return the same elements (the `None` element included),
    and the writer here must produce the same msgpack stream byte for byte."""
    import gzip
    import json

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    shard = os.path.join(gdir, "reference_shard.msgpack.l.gz")
    with open(os.path.join(gdir, "reference_shard.json")) as f:
        want = json.load(f)
    plain = lambda x: json.loads(json.dumps(x))  # OrderedDict / tuples -> plain JSON types
    py = list(load_msgpack_l_gz(shard, native=False))
    assert len(py) == len(want) == 6 and py[1] is None
    assert plain(py) == want
    nat = list(load_msgpack_l_gz(shard, native=True))
    assert len(nat) == 6 and nat[1] is None
    for got, ref in zip(nat, want):
        if ref is None:
            continue
        g = got["graph"]
        ga = dict(ref["graph"])
        add_open_vocab_nodes_and_edges(ga)  # the native reader adds the subtoken nodes / HasSubtoken edges while reading
        assert list(g["nodes"]) == ga["nodes"]
        assert {k: [list(e) for e in v] for k, v in g["edges"].items()} == {k: [list(e) for e in v] for k, v in ga["edges"].items()}
        assert list(g["reference_nodes"]) == list(ga["reference_nodes"])
        for k in ref:
            if k != "graph":
                assert plain(got[k]) == ref[k], k
    out = str(tmp_path / "again.msgpack.l.gz")
    save_msgpack_l_gz(py, out)
    with gzip.open(out) as a, gzip.open(shard) as b:
        assert a.read() == b.read()

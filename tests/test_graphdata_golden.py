"""CPU: row C1 at size -- `as_graph_data` / `add_open_vocab_nodes_and_edges` against what the REFERENCE's own code made
of graphs with 3 000 / 6 000 / 20 000 nodes (tests/golden/make_golden_graphdata.py; reference
buglab/representations/data.py:97-167).  On graphs this large the reference's walk over its `set` of token nodes is not
in sorted order; the subtoken node numbering and the HasSubtoken edge order must still be identical, bit for bit, on
both reader paths (Python and native)."""
import json
import os
import sys

import numpy as np
import pytest

from buglab.data import native
from buglab.representations.data import BugLabData
from buglab.utils.msgpackutils import load_msgpack_l_gz

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_graphdata import digest  # noqa: E402  (the digest the fixture was written with)


@pytest.fixture(scope="module")
def want(golden_dir):
    with open(os.path.join(golden_dir, "graphdata_large.json")) as f:
        return json.load(f)["datapoints"]


def _shard(golden_dir):
    return os.path.join(golden_dir, "reference_shard_large.msgpack.l.gz")


def _check(points, want):
    assert len(points) == len(want) == 3
    for d, w in zip(points, want):
        gd, tgt = BugLabData.as_graph_data(d)
        got = digest(gd, tgt)
        assert got["num_nodes"] == w["num_nodes"]
        assert got["has_subtoken_head"] == w["has_subtoken_head"]
        assert got["has_subtoken_tail"] == w["has_subtoken_tail"]
        assert got["edges"]["HasSubtoken"] == w["edges"]["HasSubtoken"]
        assert got == w


def test_python_reader_path_matches_reference_at_size(golden_dir, want, monkeypatch):
    monkeypatch.setenv("BUGLAB_NATIVE_READER", "0")
    points = list(load_msgpack_l_gz(_shard(golden_dir)))
    assert not isinstance(points[0]["graph"], native.NativeGraph)
    _check(points, want)


@pytest.mark.skipif(not native.available(), reason="libbuglab_data.so not built")
def test_native_reader_path_matches_reference_at_size(golden_dir, want):
    points = list(native.load_msgpack_l_gz_native(_shard(golden_dir)))
    assert isinstance(points[0]["graph"], native.NativeGraph)
    _check(points, want)


@pytest.mark.skipif(not native.available(), reason="libbuglab_data.so not built")
def test_native_set_order_is_cpythons():
    """bl_pyset_order (csrc_data/bl_data.cpp::cpython_int_set_order) against this interpreter's own `set`: random key
    ranges (dense, sparse, wrapping around the table), duplicate-heavy insertion sequences, sizes across every resize
    threshold up to the 50 000-entry growth-policy switch."""
    rng = np.random.default_rng(7)
    sizes = [0, 1, 4, 5, 6, 19, 20, 77, 78, 307, 308, 1229, 1230, 4915, 4916, 19661, 19662, 50001, 78643, 78644, 120000]
    for n in sizes:
        for hi in (max(n, 1), 3 * max(n, 1) + 7, 40 * max(n, 1) + 1000, 2 ** 31 - 1):
            keys = rng.integers(0, hi, size=n, dtype=np.int64)
            if n % 2:
                keys = np.concatenate([keys, keys[: n // 3], keys[::-1][: n // 5]])  # repeats, as NextToken endpoints have
            s = set()
            for k in keys.tolist():
                s.add(k)
            got = native.pyset_order(keys.astype(np.int32))
            assert got.tolist() == list(s), (n, hi)
    # chains like NextToken's: (a, b), (b, c), ... over scattered ids
    for n, hi in ((1000, 3000), (1080, 6000), (4400, 20000), (9000, 20000)):
        ids = rng.choice(hi, size=n, replace=False)
        seq = np.stack([ids[:-1], ids[1:]], 1).reshape(-1)
        s = set()
        for k in seq.tolist():
            s.add(k)
        assert native.pyset_order(seq).tolist() == list(s)

"""Native (C++) reader for `*.msgpack.l.gz` shards -- SURVEY.md section 8(f) rank 4.

`load_msgpack_l_gz_native(path)` yields the same datapoints as
`buglab.utils.msgpackutils.load_msgpack_l_gz` (reference buglab/utils/msgpackutils.py:11-14), except
that `datapoint["graph"]` is a `NativeGraph`: node strings stay in one byte blob, edge lists are
int32 `[E, 2]` arrays, and the subtoken nodes / `HasSubtoken` edges of
buglab/representations/data.py:97-121 are already added.  `BugLabData.as_graph_data` and
`StrElementRepresentationModel.tensorize_nodes` recognise it and skip their Python loops (subtoken
splitting + vocabulary lookup run in `bl_tensorize_nodes`); every other consumer sees the usual
mapping interface (`graph["nodes"][i]`, `graph["edges"].get("Child")`, ...), materialised lazily.

The library is host-only (g++ + zlib, `csrc_data/`); it is NOT part of the GPU hot path and nothing
here touches the HIP extension."""
import ctypes
import os
from collections import OrderedDict
from collections.abc import Mapping, Sequence
from ctypes import POINTER, Structure, c_char_p, c_int32, c_int64, c_uint8, c_void_p
from typing import Any, Dict, Iterator, List, Optional

import msgpack
import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbuglab_data.so")


class NativeDataUnavailable(RuntimeError):
    pass


class bl_datapoint_t(Structure):
    _fields_ = [
        ("is_nil", c_int32), ("num_nodes", c_int32), ("num_file_nodes", c_int32), ("created_has_subtoken", c_int32),
        ("node_text", c_void_p), ("node_text_off", POINTER(c_int32)),
        ("num_edge_kinds", c_int32), ("edge_kind", POINTER(c_char_p)), ("edge_pairs", POINTER(POINTER(c_int32))),
        ("edge_feat", POINTER(POINTER(c_int32))), ("edge_count", POINTER(c_int32)),
        ("feat_text", c_void_p), ("feat_text_off", POINTER(c_int32)), ("num_feats", c_int32),
        ("reference_nodes", POINTER(c_int32)), ("num_reference_nodes", c_int32),
        ("non_ascii_identifier", c_int32), ("rest", POINTER(c_uint8)), ("rest_len", c_int64),
    ]


class bl_graph_in_t(Structure):
    _fields_ = [("num_nodes", c_int32), ("token_stride", c_int32), ("token_ids", c_void_p), ("token_lens", c_void_p),
                ("adj", c_void_p), ("adj_count", c_void_p), ("adj_feat", c_void_p)]


class bl_collated_t(Structure):
    _fields_ = [("num_nodes", c_int64), ("num_messages", c_int64), ("token_ids", c_void_p), ("token_lens", c_void_p),
                ("msg_src", c_void_p), ("msg_tgt", c_void_p), ("type_ptr", c_void_p), ("tgt_ptr", c_void_p), ("tgt_msgs", c_void_p),
                ("src_ptr", c_void_p), ("src_msgs", c_void_p), ("node_order", c_void_p), ("occ_capacity", c_int64),
                ("tok_occ", c_void_p), ("tok_chunk_ptr", c_void_p), ("tok_chunk_id", c_void_p), ("num_occ", c_int64), ("num_chunks", c_int64),
                ("msg_feat", c_void_p)]


_SIGNATURES = {
    "bl_data_last_error": ([], c_char_p),
    "bl_data_version": ([], c_int32),
    "bl_reader_open": ([c_char_p], c_void_p),
    "bl_reader_next": ([c_void_p, POINTER(bl_datapoint_t)], c_int32),
    "bl_reader_close": ([c_void_p], None),
    "bl_pyset_order": ([c_void_p, c_int32, c_void_p], c_int32),
    "bl_vocab_create": ([c_char_p, c_void_p, c_int32], c_void_p),
    "bl_vocab_free": ([c_void_p], None),
    "bl_tensorize_nodes": ([c_void_p, c_int32, c_char_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p], c_int32),
    "bl_counting_sort": ([c_void_p, c_int64, c_int32, c_void_p, c_void_p], c_int32),
    "bl_collate_graphs": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(bl_collated_t)], c_int32),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeDataUnavailable(f"{LIB_PATH} not found: build it with `make -C neurips21-self-supervised-bug-detection-and-repair_amd/csrc_data`")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
        _lib = lib
    return _lib


def available() -> bool:
    return os.path.exists(LIB_PATH)


def pyset_order(inserted) -> np.ndarray:
    """Iteration order of a CPython set after inserting the given non-negative ints in order (bl_pyset_order)."""
    a = np.ascontiguousarray(inserted, dtype=np.int32)
    out = np.empty(a.shape[0], dtype=np.int32)
    lib = load_library()
    n = lib.bl_pyset_order(a.ctypes.data, int(a.shape[0]), out.ctypes.data)
    if n < 0:
        raise ValueError(lib.bl_data_last_error().decode())
    return out[:n]


def counting_sort(keys: np.ndarray, num_keys: int):
    """Stable sort of integer `keys` in [0, num_keys): -> (ptr int32 [num_keys + 1], perm int32 [E]).  Native when the
    library is built, NumPy otherwise (same result)."""
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    E = int(keys.shape[0])
    if available():
        lib = load_library()
        ptr = np.empty(num_keys + 1, dtype=np.int32)
        perm = np.empty(E, dtype=np.int32)
        rc = lib.bl_counting_sort(keys.ctypes.data, E, int(num_keys), ptr.ctypes.data, perm.ctypes.data)
        if rc != 0:
            raise ValueError(lib.bl_data_last_error().decode())
        return ptr, perm
    ptr = np.zeros(num_keys + 1, dtype=np.int32)
    if E:
        np.cumsum(np.bincount(keys, minlength=num_keys), out=ptr[1:])
    return ptr, np.argsort(keys, kind="stable").astype(np.int32)


def collate_graph_arrays(graphs, num_edge_types: int, hub_degree: int, token_chunk: int) -> Dict[str, np.ndarray]:
    """The node / message / CSR / order / token-chunk arrays of `buglab.data.collate.collate_graphs`, built by
    `bl_collate_graphs` in one GIL-free call.  `graphs`: TensorizedGraphData (token_ids, token_lens, adjacency_lists)."""
    lib = load_library()
    B, T = len(graphs), int(num_edge_types)
    S = max((g.token_ids.shape[1] for g in graphs), default=1)
    gin = (bl_graph_in_t * max(B, 1))()
    keep = []  # arrays referenced by raw pointers must outlive the call
    N = E = 0
    empty = np.zeros((0, 2), dtype=np.int32)
    with_feat = any(getattr(g, "edge_feature_ids", None) is not None for g in graphs)  # per-edge feature-token ids (edge features on)
    if with_feat and lib.bl_data_version() < 3:
        raise RuntimeError("libbuglab_data.so predates the per-edge payload of bl_collate_graphs: rebuild it")
    for b, g in enumerate(graphs):
        ids = np.ascontiguousarray(g.token_ids, dtype=np.int32)
        lens = np.ascontiguousarray(g.token_lens, dtype=np.int32)
        lists = [np.ascontiguousarray(g.adjacency_lists[t] if t < len(g.adjacency_lists) else empty, dtype=np.int32).reshape(-1, 2)
                 for t in range(T)]
        counts = np.array([a.shape[0] for a in lists], dtype=np.int32)
        ptrs = (c_void_p * T)(*[a.ctypes.data for a in lists])
        keep.append((ids, lens, lists, counts, ptrs))
        gin[b].num_nodes, gin[b].token_stride = ids.shape[0], max(1, ids.shape[1])
        gin[b].token_ids, gin[b].token_lens = ids.ctypes.data, lens.ctypes.data
        gin[b].adj, gin[b].adj_count = ctypes.cast(ptrs, c_void_p), counts.ctypes.data
        gin[b].adj_feat = None
        if with_feat:
            if g.edge_feature_ids is None:
                raise ValueError("every graph of the minibatch needs one feature id per edge")
            feats = [np.ascontiguousarray(g.edge_feature_ids[t] if t < len(g.edge_feature_ids) else np.zeros(0, np.int32), dtype=np.int32).reshape(-1)
                     for t in range(T)]
            if [f.shape[0] for f in feats] != counts.tolist():
                raise ValueError("every graph of the minibatch needs one feature id per edge")
            fptrs = (c_void_p * T)(*[f.ctypes.data for f in feats])
            keep.append((feats, fptrs))
            gin[b].adj_feat = ctypes.cast(fptrs, c_void_p)
        N += ids.shape[0]
        E += int(counts.sum())
    cap = N * S
    o = {"token_ids": np.empty((N, S), np.int32), "token_lens": np.empty(N, np.int32), "msg_src": np.empty(E, np.int32),
         "msg_tgt": np.empty(E, np.int32), "type_ptr": np.empty(T + 1, np.int32), "tgt_ptr": np.empty(N + 1, np.int32),
         "tgt_msgs": np.empty(E, np.int32), "src_ptr": np.empty(N + 1, np.int32), "src_msgs": np.empty(E, np.int32),
         "node_order": np.empty(N, np.int32), "tok_occ": np.empty(cap, np.int32), "tok_chunk_ptr": np.empty(cap + 1, np.int32),
         "tok_chunk_id": np.empty(cap, np.int32)}
    if with_feat:
        o["msg_feat"] = np.empty(E, np.int32)
    c = bl_collated_t()
    c.msg_feat = None
    c.num_nodes, c.num_messages, c.occ_capacity = N, E, cap
    for k, a in o.items():
        setattr(c, k, a.ctypes.data)
    rc = lib.bl_collate_graphs(ctypes.cast(gin, c_void_p), B, T, S, int(hub_degree), int(token_chunk), ctypes.byref(c))
    if rc != 0:
        raise ValueError(lib.bl_data_last_error().decode())
    o["tok_occ"] = o["tok_occ"][: c.num_occ]
    o["tok_chunk_ptr"] = o["tok_chunk_ptr"][: c.num_chunks + 1]
    o["tok_chunk_id"] = o["tok_chunk_id"][: c.num_chunks]
    return o


def _copy_i32(ptr, n: int) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=np.int32)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class NativeNodes(Sequence):
    """Node strings of one graph: a UTF-8 blob + offsets; `str`s are created only when asked for."""

    def __init__(self, blob: bytes, off: np.ndarray):
        self.blob, self.off = blob, off
        self._list: Optional[List[str]] = None

    def __len__(self) -> int:
        return int(self.off.shape[0]) - 1

    def as_list(self) -> List[str]:
        if self._list is None:
            b, o = self.blob, self.off.tolist()
            self._list = [b[o[i]:o[i + 1]].decode("utf-8") for i in range(len(o) - 1)]
        return self._list

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.as_list()[i]
        if self._list is not None:
            return self._list[i]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self.blob[int(self.off[i]):int(self.off[i + 1])].decode("utf-8")

    def __iter__(self):
        return iter(self.as_list())

    def __eq__(self, other):
        return isinstance(other, (list, tuple, NativeNodes)) and list(self) == list(other)

    __hash__ = None


class NativeEdges(Mapping):
    """`graph["edges"]`: edge kind -> list of `[src, tgt]` / `[src, tgt, feature]`, built on demand from
    the int32 arrays (`arrays[kind]` is what the tensoriser uses directly)."""

    def __init__(self, arrays: "OrderedDict[str, np.ndarray]", feats: Dict[str, np.ndarray], feat_strings: List[str]):
        self.arrays, self._feats, self._feat_strings = arrays, feats, feat_strings
        self._lists: Dict[str, list] = {}

    def __getitem__(self, kind):
        if kind not in self.arrays:
            raise KeyError(kind)
        out = self._lists.get(kind)
        if out is None:
            out = self.arrays[kind].tolist()
            f = self._feats.get(kind)
            if f is not None:
                for i in np.flatnonzero(f >= 0).tolist():
                    out[i].append(self._feat_strings[int(f[i])])
            self._lists[kind] = out
        return out

    def __iter__(self):
        return iter(self.arrays)

    def __len__(self):
        return len(self.arrays)

    def labelled(self, kind: str, label: str) -> List[List[int]]:
        """[src, tgt] of the edges of `kind` whose third element is `label`, in file order."""
        f = self._feats.get(kind)
        if f is None:
            return []
        wanted = [i for i, s in enumerate(self._feat_strings) if s == label]
        if not wanted:
            return []
        return self.arrays[kind][np.isin(f, wanted)].tolist()


class NativeGraph(Mapping):
    def __init__(self, nodes: NativeNodes, edges: NativeEdges, reference_nodes: np.ndarray, other: Dict[str, Any]):
        self.nodes, self.edges, self.reference_nodes_array, self._other = nodes, edges, reference_nodes, other
        self._refs_list: Optional[List[int]] = None

    def __getitem__(self, key):
        if key == "nodes":
            return self.nodes
        if key == "edges":
            return self.edges
        if key == "reference_nodes":
            if self._refs_list is None:
                self._refs_list = self.reference_nodes_array.tolist()
            return self._refs_list
        return self._other[key]

    def __iter__(self):
        yield from ("nodes", "edges", "reference_nodes")
        yield from self._other

    def __len__(self):
        return 3 + len(self._other)


def _python_fallback_graph(dp: bl_datapoint_t, nodes: NativeNodes, edges: NativeEdges, refs: np.ndarray, other) -> Dict[str, Any]:
    """An identifier token holds non-ASCII characters: redo the open-vocabulary step in Python (Unicode
    lower-casing), from the nodes / edges exactly as they are in the file."""
    from buglab.representations.data import add_open_vocab_nodes_and_edges

    g = OrderedDict(other)
    g["nodes"] = nodes.as_list()[: dp.num_file_nodes]
    kinds = list(edges.arrays)
    if dp.created_has_subtoken:
        kinds = kinds[:-1]
    g["edges"] = OrderedDict((k, edges[k]) for k in kinds)
    g["reference_nodes"] = refs.tolist()
    add_open_vocab_nodes_and_edges(g)
    return g


def load_msgpack_l_gz_native(filename) -> Iterator[Any]:
    lib = load_library()
    reader = lib.bl_reader_open(os.fsencode(str(filename)))
    if not reader:
        raise OSError(lib.bl_data_last_error().decode())
    dp = bl_datapoint_t()
    try:
        while True:
            rc = lib.bl_reader_next(reader, ctypes.byref(dp))
            if rc == 0:
                return
            if rc < 0:
                raise ValueError(f"{filename}: {lib.bl_data_last_error().decode()}")
            if dp.is_nil:
                yield None
                continue
            n = dp.num_nodes
            off = _copy_i32(dp.node_text_off, n + 1)
            blob = ctypes.string_at(dp.node_text, int(off[-1])) if n else b""
            nodes = NativeNodes(blob, off)
            foff = _copy_i32(dp.feat_text_off, dp.num_feats + 1)
            fblob = ctypes.string_at(dp.feat_text, int(foff[-1])) if dp.num_feats else b""
            feat_strings = [fblob[foff[i]:foff[i + 1]].decode("utf-8") for i in range(dp.num_feats)]
            arrays: "OrderedDict[str, np.ndarray]" = OrderedDict()
            feats: Dict[str, np.ndarray] = {}
            unusual_feature = False
            for k in range(dp.num_edge_kinds):
                kind = dp.edge_kind[k].decode("utf-8")
                cnt = dp.edge_count[k]
                arrays[kind] = _copy_i32(dp.edge_pairs[k], 2 * cnt).reshape(cnt, 2)
                f = _copy_i32(dp.edge_feat[k], cnt)
                if (f != -1).any():
                    feats[kind] = f
                    unusual_feature |= bool((f == -2).any())
            if unusual_feature:
                raise ValueError(f"{filename}: an edge carries a non-string third element; use the Python reader for this shard")
            refs = _copy_i32(dp.reference_nodes, dp.num_reference_nodes)
            rest = msgpack.unpackb(ctypes.string_at(dp.rest, dp.rest_len), raw=False, object_pairs_hook=OrderedDict, strict_map_key=False)
            other = rest.pop("graph")
            edges = NativeEdges(arrays, feats, feat_strings)
            if dp.non_ascii_identifier:
                rest["graph"] = _python_fallback_graph(dp, nodes, edges, refs, other)
            else:
                rest["graph"] = NativeGraph(nodes, edges, refs, other)
            yield rest
    finally:
        lib.bl_reader_close(reader)


class NativeVocabulary:
    """Subtoken -> id table on the C++ side (built once per model from its `Vocabulary`)."""

    def __init__(self, id_to_token: Sequence, unk_id: int):
        lib = load_library()
        enc = [t.encode("utf-8") for t in id_to_token]
        off = np.zeros(len(enc) + 1, dtype=np.int32)
        np.cumsum([len(e) for e in enc], out=off[1:])
        self._blob = b"".join(enc)
        self._off = off
        self.unk_id = int(unk_id)
        self._handle = lib.bl_vocab_create(self._blob, off.ctypes.data, len(enc))
        if not self._handle:
            raise RuntimeError(lib.bl_data_last_error().decode())

    def tensorize(self, nodes: NativeNodes, max_subtokens: int):
        """-> ids int32 [n, S], lens int32 [n], needs_python bool [n] (strings with non-ASCII characters)."""
        lib = load_library()
        n = len(nodes)
        ids = np.zeros((n, max_subtokens), dtype=np.int32)
        lens = np.ones(n, dtype=np.int32)
        needs = np.zeros(n, dtype=np.uint8)
        if n:
            rc = lib.bl_tensorize_nodes(self._handle, self.unk_id, nodes.blob, nodes.off.ctypes.data, n, max_subtokens,
                                        ids.ctypes.data, lens.ctypes.data, needs.ctypes.data)
            if rc != 0:
                raise RuntimeError(lib.bl_data_last_error().decode())
        return ids, lens, needs.astype(bool)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and _lib is not None:
                _lib.bl_vocab_free(self._handle)
        except Exception:
            pass

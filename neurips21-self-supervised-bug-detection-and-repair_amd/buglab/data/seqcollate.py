"""Host-side arrays of the sequence models' minibatches (reference buglab/models/seqmodel.py:770-975 builds
`edges [E, 3] = (sample, source token, target token)` and `edge_types [E]` with Python list appends)."""
from __future__ import annotations

import numpy as np

I32 = np.int32


def edge_csr(edges: np.ndarray, edge_types: np.ndarray, B: int, L: int):
    """Edges of a padded [B, L] minibatch -> CSR over QUERY rows (b * L + i), the form the relational attention
    kernels read (csrc/bl_seq_ops.hip).  Every edge (s, src, tgt, t) yields two entries
    (relational_multihead_attention.py:90-112): at query row (s, src): (key = tgt, code = 2 t)      -- forward bias
                                                  at query row (s, tgt): (key = src, code = 2 t + 1)  -- reverse bias
    -> row_ptr int32 [B * L + 1], key int32 [2 E], code int32 [2 E]; entries of a row keep edge-list order."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 3)
    t = np.asarray(edge_types, dtype=np.int64).reshape(-1)
    assert edges.shape[0] == t.shape[0]
    if edges.shape[0] == 0:
        return np.zeros(B * L + 1, dtype=I32), np.zeros(0, dtype=I32), np.zeros(0, dtype=I32)
    s, src, tgt = edges[:, 0], edges[:, 1], edges[:, 2]
    assert (s >= 0).all() and (s < B).all() and (src >= 0).all() and (src < L).all() and (tgt >= 0).all() and (tgt < L).all()
    rows = np.concatenate([s * L + src, s * L + tgt])
    keys = np.concatenate([tgt, src])
    codes = np.concatenate([2 * t, 2 * t + 1])
    order = np.argsort(rows, kind="stable")
    row_ptr = np.zeros(B * L + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=B * L), out=row_ptr[1:])
    return row_ptr.astype(I32), keys[order].astype(I32), codes[order].astype(I32)

"""Relational transformer encoder layer of the sequence models (`seq-great`, `seq-rat`) -- MI355X counterpart of
reference buglab/models/layers/relational_transformer.py (+ relational_multihead_attention.py,
multihead_attention.py).  Same constructor kwargs and `forward(src, src_mask, edges, edge_types)` meaning; the
arithmetic runs in libbuglab_hip: projections and the attention products on the MFMA GEMMs, edge terms / masked
softmax / LayerNorm in csrc/bl_seq_ops.hip.

Kept quirks of the reference (oracle/great_oracle.py pins them to vectors from the reference's own layers):
  * queries are scaled by dk^-0.5 BEFORE both the Q.K^T product and the edge terms (multihead_attention.py:54);
  * `seq-great` never sets `edge_attention_bias_is_scalar`, so it runs the vector QUERY-bias branch
    (relational_multihead_attention.py:135-152), not the scalar key bias of the GREAT paper;
  * under "postnorm" the second sublayer is normalised with `norm1` again; `norm2` exists but is never used
    (relational_transformer.py:123-124);
  * dropout sits on the attention probabilities, on both sublayer outputs and inside the feed-forward block.
Differences in mechanism only: weights are stored [in, out]; dropout masks come from the library's counter hash
(stateless, reproducible in backward) instead of torch's Philox stream; activations are 2-D [B * L, D].
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from buglab.models import hip_ops
from buglab.models.hip_ops import Dropout, RelEdges


def _u(shape, bound):
    return nn.Parameter(torch.empty(shape).uniform_(-bound, bound))


class RelationalTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model: int, key_query_dimension: int, value_dimension: int, nhead: int, num_edge_types: int,
                 dim_feedforward: int = 2048, dropout: float = 0.1, activation: str = "relu",
                 use_edge_value_biases: bool = False, edge_attention_bias_is_scalar: bool = False,
                 rezero_mode: str = "off", normalisation_mode: str = "postnorm"):
        super().__init__()
        if key_query_dimension != value_dimension:
            raise NotImplementedError("the HIP attention path keeps key/query and value head dimensions equal (seqmodel.py:95-96 always does)")
        if activation != "relu":
            raise NotImplementedError("feed-forward activation: relu (the reference default; seqmodel.py never overrides it)")
        if rezero_mode not in ("off", "scalar", "vector"):
            raise ValueError(f"Unrecognized rezero mode `{rezero_mode}`.")
        if normalisation_mode not in ("off", "prenorm", "postnorm"):
            raise ValueError(f"Unrecognized normalization mode `{normalisation_mode}`.")
        D, H, dk, T, FF = d_model, nhead, key_query_dimension, num_edge_types, dim_feedforward
        self.d_model, self.nhead, self.head_dim, self.num_edge_types = D, H, dk, T
        self.dropout_rate = dropout
        self._normalisation_mode, self._rezero_mode = normalisation_mode, rezero_mode
        self._scalar_bias, self._value_biases = edge_attention_bias_is_scalar, use_edge_value_biases
        # nn.Linear(D, H * 3 dk, bias=False): per head [q | k | v]   (multihead_attention.py:27-31)
        self.qkv_W = _u((D, 3 * H * dk), 1.0 / math.sqrt(D))
        self.out_W = _u((H * dk, D), 1.0 / math.sqrt(H * dk))  # _out_proj, bias=False (:35)
        bdim = H if edge_attention_bias_is_scalar else H * dk
        self.edge_bias_f = nn.Parameter(torch.randn(T, bdim))  # nn.Embedding init (relational_multihead_attention.py:59-62)
        self.edge_bias_r = nn.Parameter(torch.randn(T, bdim))
        if use_edge_value_biases:
            self.edge_vbias_f = nn.Parameter(torch.randn(T, H * dk))
            self.edge_vbias_r = nn.Parameter(torch.randn(T, H * dk))
        else:
            self.edge_vbias_f = self.edge_vbias_r = None
        self.lin1_W, self.lin1_b = _u((D, FF), 1.0 / math.sqrt(D)), _u((FF,), 1.0 / math.sqrt(D))
        self.lin2_W, self.lin2_b = _u((FF, D), 1.0 / math.sqrt(FF)), _u((D,), 1.0 / math.sqrt(FF))
        if normalisation_mode in ("prenorm", "postnorm"):
            self.norm1_g, self.norm1_b = nn.Parameter(torch.ones(D)), nn.Parameter(torch.zeros(D))
            self.norm2_g, self.norm2_b = nn.Parameter(torch.ones(D)), nn.Parameter(torch.zeros(D))
        if rezero_mode == "scalar":
            self.alpha1, self.alpha2 = nn.Parameter(torch.tensor(0.0)), nn.Parameter(torch.tensor(0.0))
        elif rezero_mode == "vector":
            self.alpha1, self.alpha2 = nn.Parameter(torch.zeros(D)), nn.Parameter(torch.zeros(D))

    def fused_call_ok(self, B: int, L: int) -> bool:
        """Whether this layer's configuration and the minibatch shape go through the one-call-per-direction form
        (hip_ops.great_layer, csrc/bl_great_layer.hip): what `seq-great` runs -- postnorm, rezero off, vector query bias, no
        edge value biases; head dimension 32.  Everything else takes the op-by-op path below (same arithmetic)."""
        return (self._normalisation_mode == "postnorm" and self._rezero_mode == "off" and not self._scalar_bias and not self._value_biases
                and hip_ops.great_layer_ok(B, L, self.nhead, self.head_dim, self.num_edge_types, self.lin1_W.shape[1]))

    def forward(self, x: torch.Tensor, lens: torch.Tensor, edges: RelEdges, B: int, L: int,
                dropout_seed: Optional[int] = None, dropout_stream: int = 0, chain: Optional[dict] = None) -> torch.Tensor:
        """x [B * L, D]; lens int32 [B] = number of real tokens per sample (keys at positions >= lens are the
        reference's `src_mask`); edges = CSR of (sample, source, target, type) over query rows.  chain: a dict the encoder
        shares between its layers (the packed form of the activations travels in it from one fused layer call to the next)."""
        p = self.dropout_rate if (self.training and dropout_seed is not None) else 0.0
        mk = lambda site: Dropout(p, int(dropout_seed or 0), dropout_stream + site)
        post, pre = self._normalisation_mode == "postnorm", self._normalisation_mode == "prenorm"
        if x.is_cuda and self.fused_call_ok(B, L):
            return hip_ops.great_layer(x, self.qkv_W, self.out_W, self.edge_bias_f, self.edge_bias_r, self.lin1_W, self.lin1_b, self.lin2_W,
                                       self.lin2_b, self.norm1_g, self.norm1_b, lens, edges, B, L, self.nhead, self.head_dim,
                                       self.num_edge_types, drops=(mk(0), mk(1), mk(2), mk(3)), chain=chain)
        # --- sublayer 1: relational self-attention (relational_transformer.py:104-113)
        a_in = hip_ops.add_layernorm(x, None, self.norm1_g, self.norm1_b) if pre else x
        qkv = hip_ops.gather_linear([(a_in, None)], self.qkv_W, None)
        ctx = hip_ops.rel_attention(qkv, lens, edges, self.edge_bias_f, self.edge_bias_r, self.edge_vbias_f, self.edge_vbias_r,
                                    B, L, self.nhead, self.head_dim, self.num_edge_types, self._scalar_bias, mk(0))
        if self._rezero_mode == "off":
            att = hip_ops.gather_linear([(ctx, None)], self.out_W, None, drop=mk(1))
        else:
            att = hip_ops.dropout_rows(hip_ops.gather_linear([(ctx, None)], self.out_W, None) * self.alpha1, mk(1))
        x = hip_ops.add_layernorm(x, att, self.norm1_g, self.norm1_b) if post else x + att
        # --- sublayer 2: feed-forward (:115-124)
        f_in = hip_ops.add_layernorm(x, None, self.norm2_g, self.norm2_b) if pre else x
        hidden = hip_ops.gather_linear([(f_in, None)], self.lin1_W, self.lin1_b, "relu", drop=mk(2))
        if self._rezero_mode == "off":
            ff = hip_ops.gather_linear([(hidden, None)], self.lin2_W, self.lin2_b, drop=mk(3))
        else:
            ff = hip_ops.dropout_rows(hip_ops.gather_linear([(hidden, None)], self.lin2_W, self.lin2_b) * self.alpha2, mk(3))
        # sic: norm1 again -- the reference normalises the second sublayer with norm1 (:123-124)
        return hip_ops.add_layernorm(x, ff, self.norm1_g, self.norm1_b) if post else x + ff

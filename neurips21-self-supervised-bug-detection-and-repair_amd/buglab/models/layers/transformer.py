"""Plain transformer encoder layer of `seq-transformer` -- MI355X counterpart of the `torch.nn.TransformerEncoderLayer` the
reference stacks for that model (buglab/models/seqmodel.py:108-118: d_model, nhead, dim_feedforward, dropout; called at
:380-384 with `src_key_padding_mask`).  torch's defaults are what the reference runs: post-norm (`norm_first=False`), relu,
`nn.MultiheadAttention` with biased input / output projections, queries scaled by dk^-0.5 after the bias, dropout on the
attention probabilities, on both sublayer outputs and inside the feed-forward block, and -- unlike the relational layer
(relational_transformer.py:123-124) -- a second LayerNorm of its own for the second sublayer.

The arithmetic runs on the same library calls as the relational layer's op-by-op path: projections on the MFMA row GEMMs
(bias / relu / dropout in their epilogues), scores + masked softmax + dropout in the fused attention kernel with an EMPTY
edge list (no relational terms), LayerNorm fused with the residual sum.  Parameters are stored [in, out] with the QKV columns
per head [q | k | v] (the layout the attention kernels read); `load_torch_layer` / `to_torch_layer` map to and from
torch's layout, and tests/test_seq_transformer_gpu.py checks this layer against torch.nn.TransformerEncoderLayer itself.
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


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1, activation: str = "relu"):
        super().__init__()
        if d_model % nhead != 0:
            raise AssertionError("embed_dim must be divisible by num_heads")  # nn.MultiheadAttention's own assertion
        if activation != "relu":
            raise NotImplementedError("feed-forward activation: relu (torch's default; seqmodel.py:108-118 never overrides it)")
        D, H, FF = d_model, nhead, dim_feedforward
        dk = D // H
        self.d_model, self.nhead, self.head_dim, self.dropout_rate = D, H, dk, dropout
        # nn.MultiheadAttention._reset_parameters: xavier_uniform_ on the [3 D, D] in_proj_weight, zero biases
        self.in_proj_W = _u((D, 3 * H * dk), math.sqrt(6.0 / (D + 3 * D)))
        self.in_proj_b = nn.Parameter(torch.zeros(3 * H * dk))
        self.out_W = _u((H * dk, D), 1.0 / math.sqrt(D))  # nn.Linear default (kaiming_uniform a = sqrt 5)
        self.out_b = nn.Parameter(torch.zeros(D))
        self.lin1_W, self.lin1_b = _u((D, FF), 1.0 / math.sqrt(D)), _u((FF,), 1.0 / math.sqrt(D))
        self.lin2_W, self.lin2_b = _u((FF, D), 1.0 / math.sqrt(FF)), _u((D,), 1.0 / math.sqrt(FF))
        self.norm1_g, self.norm1_b = nn.Parameter(torch.ones(D)), nn.Parameter(torch.zeros(D))
        self.norm2_g, self.norm2_b = nn.Parameter(torch.ones(D)), nn.Parameter(torch.zeros(D))
        # the attention kernels take the relational terms' tables as arguments: one all-zero row each, no edges -> never read
        self.register_buffer("_no_bias", torch.zeros(1, H * dk), persistent=False)

    # -- torch layout <-> this layer's ------------------------------------------------------------------------------
    @torch.no_grad()
    def load_torch_layer(self, layer: "nn.TransformerEncoderLayer") -> "TransformerEncoderLayer":
        D, H, dk = self.d_model, self.nhead, self.head_dim
        a = layer.self_attn
        dt, dev = self.in_proj_W.dtype, self.in_proj_W.device
        c = lambda t: t.detach().to(device=dev, dtype=dt).contiguous()
        self.in_proj_W.copy_(c(a.in_proj_weight.view(3, H, dk, D).permute(3, 1, 0, 2).reshape(D, 3 * H * dk)))
        self.in_proj_b.copy_(c(a.in_proj_bias.view(3, H, dk).permute(1, 0, 2).reshape(-1)))
        self.out_W.copy_(c(a.out_proj.weight.t()))
        self.out_b.copy_(c(a.out_proj.bias))
        self.lin1_W.copy_(c(layer.linear1.weight.t()))
        self.lin1_b.copy_(c(layer.linear1.bias))
        self.lin2_W.copy_(c(layer.linear2.weight.t()))
        self.lin2_b.copy_(c(layer.linear2.bias))
        for mine, theirs in ((self.norm1_g, layer.norm1.weight), (self.norm1_b, layer.norm1.bias), (self.norm2_g, layer.norm2.weight),
                             (self.norm2_b, layer.norm2.bias)):
            mine.copy_(c(theirs))
        return self

    def torch_layout(self, tensors: Optional[dict] = None) -> dict:
        """this layer's parameters (or, given {name: tensor}, e.g. their gradients) under torch.nn.TransformerEncoderLayer's names and shapes"""
        D, H, dk = self.d_model, self.nhead, self.head_dim
        t = tensors if tensors is not None else {k: v.detach() for k, v in self.named_parameters()}
        return {
            "self_attn.in_proj_weight": t["in_proj_W"].view(D, H, 3, dk).permute(2, 1, 3, 0).reshape(3 * D, D),
            "self_attn.in_proj_bias": t["in_proj_b"].view(H, 3, dk).permute(1, 0, 2).reshape(-1),
            "self_attn.out_proj.weight": t["out_W"].t(), "self_attn.out_proj.bias": t["out_b"],
            "linear1.weight": t["lin1_W"].t(), "linear1.bias": t["lin1_b"], "linear2.weight": t["lin2_W"].t(), "linear2.bias": t["lin2_b"],
            "norm1.weight": t["norm1_g"], "norm1.bias": t["norm1_b"], "norm2.weight": t["norm2_g"], "norm2.bias": t["norm2_b"],
        }

    def forward(self, x: torch.Tensor, lens: torch.Tensor, edges: Optional[RelEdges], B: int, L: int,
                dropout_seed: Optional[int] = None, dropout_stream: int = 0, chain: Optional[dict] = None) -> torch.Tensor:
        """x [B * L, D]; lens int32 [B]: keys at positions >= lens are masked (`src_key_padding_mask`); `edges` is ignored
        (same call shape as RelationalTransformerEncoderLayer.forward)."""
        p = self.dropout_rate if (self.training and dropout_seed is not None) else 0.0
        mk = lambda site: Dropout(p, int(dropout_seed or 0), dropout_stream + site)
        none = RelEdges(lens, lens, lens, 0)  # (num_entries == 0: the index arrays are never dereferenced)
        qkv = hip_ops.gather_linear([(x, None)], self.in_proj_W, self.in_proj_b)
        ctx = hip_ops.rel_attention(qkv, lens, none, self._no_bias, self._no_bias, None, None, B, L, self.nhead, self.head_dim, 1, False, mk(0))
        att = hip_ops.gather_linear([(ctx, None)], self.out_W, self.out_b, drop=mk(1))
        x = hip_ops.add_layernorm(x, att, self.norm1_g, self.norm1_b)
        hidden = hip_ops.gather_linear([(x, None)], self.lin1_W, self.lin1_b, "relu", drop=mk(2))
        ff = hip_ops.gather_linear([(hidden, None)], self.lin2_W, self.lin2_b, drop=mk(3))
        return hip_ops.add_layernorm(x, ff, self.norm2_g, self.norm2_b)

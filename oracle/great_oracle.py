"""CPU oracle for the `seq-great` relational-transformer block (SURVEY.md section 8f rank 1,
BASELINE.json configs[4]).  TEST INFRASTRUCTURE ONLY -- groundwork for the device path of a later round:
no product code imports this file and the registry still raises NotImplementedError for "seq-great".

PARITY STATUS: **pinned**.  Unlike the ptgnn layers of `gnn-mlp`, this block lives in the reference tree
(`buglab/models/layers/multihead_attention.py`, `relational_multihead_attention.py`,
`relational_transformer.py`), is pure PyTorch and importable offline: `tests/golden/make_golden_great.py`
runs the reference's own `RelationalTransformerEncoderLayer` stack on seeded inputs and commits inputs,
weights, outputs and gradients; `tests/test_great_oracle_golden.py` checks this restatement against them.

Weights are passed as a dict with the reference's own state_dict names (per layer prefix `layers.{i}.`).
Dropout is off (parity runs use p = 0; the reference's stateful Philox masks cannot be reproduced)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class GreatConfig:
    """`seq-great` as the registry builds it: reference modelregistry.py:97-127 (hidden 256, 5 layers, 8 heads,
    FF 1024, postnorm, rezero off) and seqmodel.py:91-106 (key = value dim = D / heads, edge value biases only
    for layer_type "rat", `edge_attention_bias_is_scalar` left at its default False -> the query-bias branch)."""

    d_model: int = 256
    num_heads: int = 8
    num_layers: int = 5
    dim_feedforward: int = 1024
    num_edge_types: int = 8
    use_edge_value_biases: bool = False
    edge_attention_bias_is_scalar: bool = False
    normalisation_mode: str = "postnorm"
    activation: str = "relu"  # relational_transformer.py:29 default, not overridden by seqmodel.py

    @property
    def head_dim(self) -> int:
        return self.d_model // self.num_heads


def relational_attention(p: Dict[str, torch.Tensor], pre: str, x, masked, edges, edge_types, cfg: GreatConfig):
    """RelationalMultiheadAttention.forward (relational_multihead_attention.py:72-88).

    x [B, L, D]; masked bool [B, L] (True = padding key) or None; edges int64 [E, 3] = (sample, source, target);
    edge_types int64 [E]."""
    B, L, _ = x.shape
    H, dk = cfg.num_heads, cfg.head_dim
    # multihead_attention.py:46-57: one bias-free projection, per head [q | k | v], queries pre-scaled by dk^-0.5
    qkv = (x @ p[pre + "self_attn._selfatt_head_transforms.weight"].T).reshape(B, L, H, 3 * dk)
    q, k, v = torch.split(qkv, [dk, dk, dk], dim=-1)
    q = q * dk ** -0.5
    scores = torch.einsum("bkhd,bqhd->bqkh", k, q)  # :59-65   [B, query, key, head]
    if edges.shape[0] > 0:  # relational_multihead_attention.py:90-112
        s, src, tgt = edges[:, 0], edges[:, 1], edges[:, 2]
        bias = p[pre + "self_attn._edge_attention_biases.weight"][edge_types]
        bias_r = p[pre + "self_attn._reverse_edge_attention_biases.weight"][edge_types]
        if cfg.edge_attention_bias_is_scalar:  # :119-134 (GREAT as published: scalar x key, target side)
            e_f = torch.einsum("eh,ehd->eh", bias, k[s, tgt])
            e_r = torch.einsum("eh,ehd->eh", bias_r, k[s, src])
        else:  # :135-152 (what `seq-great` actually runs: vector bias x query, source side)
            e_f = torch.einsum("ehd,ehd->eh", bias.reshape(-1, H, dk), q[s, src])
            e_r = torch.einsum("ehd,ehd->eh", bias_r.reshape(-1, H, dk), q[s, tgt])
        scores = scores.contiguous().index_put((torch.cat([s, s]), torch.cat([src, tgt]), torch.cat([tgt, src])),
                                               torch.cat([e_f, e_r]), accumulate=True)
    scores = scores.transpose(2, 3)  # multihead_attention.py:67-77   [B, query, head, key]
    if masked is not None:
        scores = scores.masked_fill(masked[:, None, None, :], -math.inf)
    probs = F.softmax(scores, dim=-1)
    ctxv = torch.einsum("blhq,bqhd->blhd", probs, v)  # :79-80
    if cfg.use_edge_value_biases and edges.shape[0] > 0:  # relational_multihead_attention.py:155-178 ("rat")
        s, src, tgt = edges[:, 0], edges[:, 1], edges[:, 2]
        vb = probs[s, src, :, tgt].unsqueeze(-1) * p[pre + "self_attn._edge_value_biases.weight"][edge_types].reshape(-1, H, dk)
        vb_r = probs[s, tgt, :, src].unsqueeze(-1) * p[pre + "self_attn._reverse_edge_value_biases.weight"][edge_types].reshape(-1, H, dk)
        ctxv = ctxv.contiguous().index_put((torch.cat([s, s]), torch.cat([src, tgt])), torch.cat([vb, vb_r]), accumulate=True)
    return ctxv.reshape(B, L, H * dk) @ p[pre + "self_attn._out_proj.weight"].T  # multihead_attention.py:82-88


def encoder_layer(p: Dict[str, torch.Tensor], pre: str, x, masked, edges, edge_types, cfg: GreatConfig):
    """RelationalTransformerEncoderLayer.forward (relational_transformer.py:103-125), rezero off."""
    def ln(t, which):
        return F.layer_norm(t, (cfg.d_model,), p[pre + which + ".weight"], p[pre + which + ".bias"], 1e-5)

    act = F.relu if cfg.activation == "relu" else F.gelu
    a_in = ln(x, "norm1") if cfg.normalisation_mode == "prenorm" else x
    x = x + relational_attention(p, pre, a_in, masked, edges, edge_types, cfg)
    if cfg.normalisation_mode == "postnorm":
        x = ln(x, "norm1")
    f_in = ln(x, "norm2") if cfg.normalisation_mode == "prenorm" else x
    ff = act(f_in @ p[pre + "linear1.weight"].T + p[pre + "linear1.bias"]) @ p[pre + "linear2.weight"].T + p[pre + "linear2.bias"]
    x = x + ff
    if cfg.normalisation_mode == "postnorm":
        x = ln(x, "norm1")  # sic: the reference re-uses norm1 for the second sublayer (:123-124); norm2 stays unused
    return x


def encoder_stack(p: Dict[str, torch.Tensor], x, masked, edges, edge_types, cfg: GreatConfig, prefix: str = "layers."):
    """The `__seq_layers` loop of SeqBugLabModule (reference seqmodel.py, layer_type in {"great", "rat"})."""
    for i in range(cfg.num_layers):
        x = encoder_layer(p, f"{prefix}{i}.", x, masked, edges, edge_types, cfg)
    return x

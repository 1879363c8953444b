#!/usr/bin/env python
"""Golden vectors for the `seq-great` relational-transformer block, produced by the REFERENCE's own layers
(`/root/reference/buglab/models/layers/relational_transformer.py` and the two attention files: pure PyTorch,
importable offline).  Run in the build container only:   python tests/golden/make_golden_great.py
Writes tests/golden/great_{great,rat,scalar}.npz (inputs, state_dict, output, gradients)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from buglab.models.layers.relational_transformer import RelationalTransformerEncoderLayer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, *, use_value_bias, scalar, norm, seed, D=64, H=4, FF=96, T=5, L=23, B=3, layers=2, E=70):
    torch.manual_seed(seed)
    stack = torch.nn.ModuleList([
        RelationalTransformerEncoderLayer(d_model=D, key_query_dimension=D // H, value_dimension=D // H, nhead=H, num_edge_types=T,
                                          dim_feedforward=FF, dropout=0.0, use_edge_value_biases=use_value_bias,
                                          edge_attention_bias_is_scalar=scalar, normalisation_mode=norm)
        for _ in range(layers)])
    with torch.no_grad():  # LayerNorm affine away from (1, 0) so that the norm1 / norm2 mix-up is visible
        for l in stack:
            for n in (l.norm1, l.norm2):
                if n is not None:
                    n.weight.add_(0.3 * torch.randn_like(n.weight))
                    n.bias.add_(0.3 * torch.randn_like(n.bias))
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, L, D, generator=g, requires_grad=True)
    lens = torch.tensor([L, L - 5, L - 11])
    masked = torch.arange(L)[None, :] >= lens[:, None]
    s = torch.randint(0, B, (E,), generator=g)
    src = (torch.rand(E, generator=g) * lens[s]).long()
    tgt = (torch.rand(E, generator=g) * lens[s]).long()
    edges = torch.stack([s, src, tgt], 1)
    edges[5] = edges[4]  # a repeated edge: the accumulate=True of index_put_ matters
    types = torch.randint(0, T, (E,), generator=g)
    y = x
    for l in stack:
        y = l(y, masked, edges, types)
    w = torch.randn(B, L, D, generator=g)
    (y * w * (~masked)[:, :, None]).sum().backward()
    out = {"x": x.detach().numpy(), "masked": masked.numpy(), "edges": edges.numpy(), "edge_types": types.numpy(),
           "y": y.detach().numpy(), "w": w.numpy(), "g_x": x.grad.numpy(),
           "cfg": np.array([D, H, layers, FF, T, int(use_value_bias), int(scalar)]), "norm": np.array(norm)}
    for k, v in stack.state_dict().items():
        out["p." + k] = v.numpy()
    for k, v in stack.named_parameters():
        out["g." + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    np.savez_compressed(os.path.join(HERE, f"great_{name}.npz"), **out)
    print(name, float(y.abs().mean()))


CASES = {
    "great": dict(use_value_bias=False, scalar=False, norm="postnorm", seed=0),   # what the registry's seq-great runs
    "rat": dict(use_value_bias=True, scalar=False, norm="prenorm", seed=1),       # seq-rat + the prenorm branch
    "scalar": dict(use_value_bias=False, scalar=True, norm="postnorm", seed=2),   # GREAT as published (scalar key bias)
    # seq-great's configuration at head dimension 32 (BASELINE configs[4]: 256 / 8 heads): the shape the one-call-per-layer
    # form (bl_great_layer_fwd / _bwd) takes -- added in round 6; the three cases above were not regenerated
    "great32": dict(use_value_bias=False, scalar=False, norm="postnorm", seed=3, D=64, H=2, FF=96),
}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CASES)):
        make(name, **CASES[name])

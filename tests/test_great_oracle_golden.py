"""CPU: the `seq-great` oracle (oracle/great_oracle.py) against vectors produced by the reference's own
RelationalTransformerEncoderLayer (tests/golden/make_golden_great.py) -- pinned parity for the SURVEY 8(f)
rank-1 block, ahead of its device path."""
import os

import numpy as np
import pytest
import torch

from oracle import great_oracle as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["great", "rat", "scalar", "great32"])
def test_encoder_stack_matches_reference(name):
    z = np.load(os.path.join(GOLD, f"great_{name}.npz"))
    D, H, layers, FF, T, value_bias, scalar = (int(v) for v in z["cfg"])
    cfg = G.GreatConfig(d_model=D, num_heads=H, num_layers=layers, dim_feedforward=FF, num_edge_types=T,
                        use_edge_value_biases=bool(value_bias), edge_attention_bias_is_scalar=bool(scalar),
                        normalisation_mode=str(z["norm"]))
    p = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("p.")}
    x = torch.from_numpy(z["x"]).clone().requires_grad_(True)
    masked = torch.from_numpy(z["masked"])
    y = G.encoder_stack(p, x, masked, torch.from_numpy(z["edges"]), torch.from_numpy(z["edge_types"]), cfg, prefix="")
    valid = ~masked
    assert (y.detach() - torch.from_numpy(z["y"]))[valid].abs().max() < 2e-5
    (y * torch.from_numpy(z["w"]) * valid[:, :, None]).sum().backward()
    assert (x.grad - torch.from_numpy(z["g_x"])).abs().max() < 1e-4
    for k, v in p.items():
        ref = torch.from_numpy(z["g." + k])
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert (got - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max())), k
    # the reference's quirk is reproduced: under postnorm norm2 never receives a gradient
    if str(z["norm"]) == "postnorm":
        assert float(np.abs(z["g.0.norm2.weight"]).max()) == 0.0 and p["0.norm2.weight"].grad is None


def test_attention_rows_are_distributions_and_padding_is_ignored():
    torch.manual_seed(0)
    cfg = G.GreatConfig(d_model=32, num_heads=4, num_layers=1, dim_feedforward=48, num_edge_types=3)
    z = np.load(os.path.join(GOLD, "great_great.npz"))
    B, L = 2, 9
    p = {"0.self_attn._selfatt_head_transforms.weight": torch.randn(3 * 32, 32) * 0.2,
         "0.self_attn._out_proj.weight": torch.randn(32, 32) * 0.2,
         "0.self_attn._edge_attention_biases.weight": torch.randn(3, 32), "0.self_attn._reverse_edge_attention_biases.weight": torch.randn(3, 32)}
    x = torch.randn(B, L, 32)
    masked = torch.zeros(B, L, dtype=torch.bool)
    masked[1, 6:] = True
    edges = torch.tensor([[0, 1, 2], [1, 0, 5], [1, 3, 3]])
    types = torch.tensor([0, 2, 1])
    a = G.relational_attention(p, "0.", x, masked, edges, types, cfg)
    x2 = x.clone()
    x2[1, 6:] += 100.0  # padded key positions must not influence the valid query rows
    b = G.relational_attention(p, "0.", x2, masked, edges, types, cfg)
    assert (a[1, :6] - b[1, :6]).abs().max() < 1e-5 and (a[0] - b[0]).abs().max() < 1e-6
    assert z["y"].shape[0] == 3

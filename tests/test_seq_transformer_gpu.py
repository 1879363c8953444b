"""GPU: `seq-transformer` (reference seqmodel.py:108-118, 380-384) -- the HIP layer against torch.nn.TransformerEncoderLayer ITSELF,
the module the reference stacks for this model, evaluated live in float64 on the CPU with the same weights (mapped by
TransformerEncoderLayer.load_torch_layer): outputs at real positions and ALL gradients within 1e-4; then the model through
the registry name, a few optimiser steps, predict()."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from buglab.models import hip_ops

    hip_ops.load_library()


@pytest.mark.parametrize("B,L,D,H,FF,layers", [(3, 40, 64, 4, 96, 2), (2, 132, 128, 4, 256, 3), (2, 64, 64, 2, 128, 1)])
def test_stack_matches_torch_transformer_encoder_layer(B, L, D, H, FF, layers):
    from buglab.models.layers.transformer import TransformerEncoderLayer

    torch.manual_seed(B * 1000 + L)
    ref = torch.nn.ModuleList([torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=FF, dropout=0.0) for _ in range(layers)]).double()
    with torch.no_grad():  # (torch zero-initialises the projection biases: give them values so that their handling is tested)
        for l in ref:
            for b in (l.self_attn.in_proj_bias, l.self_attn.out_proj.bias):
                b.uniform_(-0.3, 0.3)
            for n in (l.norm1, l.norm2):
                n.weight.uniform_(0.5, 1.5)
                n.bias.uniform_(-0.2, 0.2)
    mine = torch.nn.ModuleList([TransformerEncoderLayer(D, H, FF, dropout=0.0).load_torch_layer(l) for l in ref]).cuda().train()
    lens = torch.tensor([L, max(1, L // 2), max(1, L - 7)][:B], dtype=torch.int32)
    x = torch.randn(B, L, D, dtype=torch.float64)
    w = torch.randn(B, L, D, dtype=torch.float64)  # loss weights
    valid = (torch.arange(L)[None, :] < lens[:, None])

    # reference: seqmodel.py:380-384 -- [L, B, D] in, src_key_padding_mask = padding positions
    xr = x.clone().requires_grad_(True)
    h = xr
    for l in ref:
        h = l(h.transpose(0, 1), src_key_padding_mask=~valid).transpose(0, 1)
    want = h
    (want * w * valid[..., None]).sum().backward()

    xm = x.float().cuda().reshape(B * L, D).requires_grad_(True)
    hm = xm
    for l in mine:
        hm = l(hm, lens.cuda(), None, B, L)
    (hm * (w * valid[..., None]).float().cuda().reshape(B * L, D)).sum().backward()
    torch.cuda.synchronize()
    got = hm.detach().cpu().double().view(B, L, D)
    assert float((got - want.detach())[valid].abs().max()) < 1e-4
    gx = xm.grad.cpu().double().view(B, L, D)
    assert float((gx - xr.grad)[valid].abs().max()) <= 1e-4 * float(xr.grad.abs().max()) + 1e-6
    for l_ref, l_mine in zip(ref, mine):
        grads = l_mine.torch_layout({k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in l_mine.named_parameters()})
        for name, p in l_ref.named_parameters():
            g = grads[name].detach().cpu().double()
            assert float((g - p.grad).abs().max()) <= 1e-4 * float(p.grad.abs().max()) + 1e-6, name
        params = l_mine.torch_layout()
        for name, p in l_ref.named_parameters():  # (and the layout maps are inverses of each other)
            assert torch.equal(params[name].cpu(), p.detach().float())


def test_seq_transformer_through_the_registry():
    from pathlib import Path

    from buglab.data.collate import to_device
    from buglab.data.synthetic import make_buglab_seq_dataset
    from buglab.models.layers.transformer import TransformerEncoderLayer
    from buglab.models.modelregistry import load_model
    from buglab.runtime.optim import FlatAdam

    data = make_buglab_seq_dataset(6, seed=5)
    model = load_model({"modelName": "seq-transformer", "hidden_state_size": 64, "num_layers": 2, "num_heads": 4, "intermediate_dimension_size": 96,
                        "dropout_rate": 0.1}, Path("/tmp/_bl_seq_tr.pkl.gz"))[0]
    model.compute_metadata(copy.deepcopy(data))
    torch.manual_seed(0)
    nn_ = model.build_neural_module().cuda().train()
    assert all(isinstance(l, TransformerEncoderLayer) for l in nn_.encoder.layers) if hasattr(nn_, "encoder") else True
    samples = [model.tensorize(copy.deepcopy(d)) for d in data]
    mb = to_device(model.collate_minibatch({"samples": samples}), "cuda")
    opt = FlatAdam(nn_.parameters(), lr=1e-3, num_warmup_steps=0)
    losses = []
    for step in range(6):
        opt.zero_grad()
        l = nn_(**mb, dropout_seed=step)
        l.backward()
        opt.step()
        losses.append(float(l.detach()))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    res = list(model.predict(iter(copy.deepcopy(data)), nn_, "cuda", parallelize=False))
    assert len(res) == len(data)
    for point, loc, rewrites in res:
        assert len(rewrites) == len(point["candidate_rewrites"]) and -1 in loc
        assert abs(sum(np.exp(v) for v in loc.values()) - 1.0) < 1e-4

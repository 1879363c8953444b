"""GPU: `seq-gru` (reference seqmodel.py:119-126, 385-392) -- the HIP bidirectional GRU stack against torch.nn.GRU ITSELF, the module
the reference runs for this model, evaluated live in float64 on the CPU over the same PackedSequence with the same weights
(BiGRULayer.load_torch_gru): outputs and ALL gradients within 1e-4; then the model through the registry name."""
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


@pytest.mark.parametrize("B,L,D,layers,lens", [(3, 24, 64, 2, [24, 9, 1]), (2, 70, 128, 3, [70, 33]), (4, 40, 256, 2, [40, 17, 40, 5])])
def test_stack_matches_torch_gru_over_packed_sequences(B, L, D, layers, lens):
    from buglab.models.layers.gru import BiGRULayer

    torch.manual_seed(L)
    gru = torch.nn.GRU(input_size=D, hidden_size=D // 2, num_layers=layers, bidirectional=True, batch_first=True).double()
    mine = torch.nn.ModuleList([BiGRULayer(D, D // 2).load_torch_gru(gru, k) for k in range(layers)]).cuda().train()
    lens_t = torch.tensor(lens, dtype=torch.int64)
    x = torch.randn(B, L, D, dtype=torch.float64)
    w = torch.randn(B, L, D, dtype=torch.float64)
    valid = torch.arange(L)[None, :] < lens_t[:, None]

    # reference: seqmodel.py:385-392
    xr = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lengths=lens_t, batch_first=True, enforce_sorted=False)
    out, _ = gru(packed)
    want, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=L)
    (want * w).sum().backward()

    xm = x.float().cuda().reshape(B * L, D).requires_grad_(True)
    hm = xm
    lens_dev = lens_t.to(torch.int32).cuda()
    for l in mine:
        hm = l(hm, lens_dev, None, B, L)
    (hm * w.float().cuda().reshape(B * L, D)).sum().backward()
    torch.cuda.synchronize()
    got = hm.detach().cpu().double().view(B, L, D)
    assert float((got - want.detach()).abs().max()) < 1e-4
    assert float(got[~valid].abs().max()) == 0.0 if (~valid).any() else True  # pad_packed_sequence's zeros
    gx = xm.grad.cpu().double().view(B, L, D)
    assert float((gx - xr.grad)[valid].abs().max()) <= 1e-4 * float(xr.grad.abs().max()) + 1e-6
    ref_grads = {n: p.grad for n, p in gru.named_parameters()}
    for k, l in enumerate(mine):
        grads = l.torch_layout(k, {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in l.named_parameters()})
        for n, g in grads.items():
            d = float((g.detach().cpu().double() - ref_grads[n]).abs().max())
            assert d <= 1e-4 * float(ref_grads[n].abs().max()) + 1e-6, (n, d)


def test_forward_only_call_keeps_nothing():
    from buglab.models.layers.gru import BiGRULayer

    torch.manual_seed(0)
    l = BiGRULayer(64, 32).cuda().eval()
    x = torch.randn(2 * 16, 64, device="cuda")
    lens = torch.tensor([16, 3], dtype=torch.int32, device="cuda")
    with torch.no_grad():
        a = l(x, lens, None, 2, 16)
    b = l(x, lens, None, 2, 16)
    assert torch.equal(a, b.detach())


def test_seq_gru_through_the_registry():
    from pathlib import Path

    from buglab.data.collate import to_device
    from buglab.data.synthetic import make_buglab_seq_dataset
    from buglab.models.modelregistry import load_model
    from buglab.runtime.optim import FlatAdam

    data = make_buglab_seq_dataset(6, seed=5)
    model = load_model({"modelName": "seq-gru", "hidden_state_size": 64, "num_layers": 2, "dropout_rate": 0.1}, Path("/tmp/_bl_seq_gru.pkl.gz"))[0]
    model.compute_metadata(copy.deepcopy(data))
    torch.manual_seed(0)
    nn_ = model.build_neural_module().cuda().train()
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

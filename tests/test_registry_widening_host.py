"""CPU: `seq-transformer` / `seq-gru` host side -- the registry builds it (reference modelregistry.py:135), the encoder stacks the plain
transformer layer, and the layer's parameter layout maps to torch.nn.TransformerEncoderLayer's and back without loss."""
import copy
from pathlib import Path

import pytest
import torch


def test_layout_maps_are_inverse():
    from buglab.models.layers.transformer import TransformerEncoderLayer

    torch.manual_seed(1)
    ref = torch.nn.TransformerEncoderLayer(d_model=64, nhead=4, dim_feedforward=96, dropout=0.0)
    with torch.no_grad():
        ref.self_attn.in_proj_bias.uniform_(-1, 1)
    mine = TransformerEncoderLayer(64, 4, 96, dropout=0.0).load_torch_layer(ref)
    back = mine.torch_layout()
    for name, p in ref.named_parameters():
        assert torch.equal(back[name], p.detach()), name
    # head h's query / key / value columns are contiguous blocks of the packed projection: x @ in_proj_W == torch's in_proj, permuted
    x = torch.randn(5, 64)
    qkv = x @ mine.in_proj_W + mine.in_proj_b
    q, k, v = torch.nn.functional.linear(x, ref.self_attn.in_proj_weight, ref.self_attn.in_proj_bias).chunk(3, dim=-1)
    for h in range(4):
        blk = qkv[:, h * 48:(h + 1) * 48]
        assert torch.allclose(blk[:, :16], q[:, h * 16:(h + 1) * 16], atol=1e-6)
        assert torch.allclose(blk[:, 16:32], k[:, h * 16:(h + 1) * 16], atol=1e-6)
        assert torch.allclose(blk[:, 32:], v[:, h * 16:(h + 1) * 16], atol=1e-6)


def test_gru_layout_maps_are_inverse():
    from buglab.models.layers.gru import BiGRULayer

    torch.manual_seed(2)
    gru = torch.nn.GRU(input_size=64, hidden_size=32, num_layers=2, bidirectional=True, batch_first=True)
    for k in range(2):
        mine = BiGRULayer(64, 32).load_torch_gru(gru, k)
        back = mine.torch_layout(k)
        want = {n: p for n, p in gru.named_parameters() if f"_l{k}" in n}
        assert set(back) == set(want)
        for n, p in want.items():
            assert torch.equal(back[n], p.detach()), n
    with pytest.raises(NotImplementedError):
        BiGRULayer(100, 50)


def test_registry_builds_seq_transformer_and_seq_gru(tmp_path):
    from buglab.data.synthetic import make_buglab_seq_dataset
    from buglab.models.layers.transformer import TransformerEncoderLayer
    from buglab.models.modelregistry import load_model

    data = make_buglab_seq_dataset(4, seed=2)
    model = load_model({"modelName": "seq-transformer", "hidden_state_size": 64, "num_layers": 3, "num_heads": 4,
                        "intermediate_dimension_size": 96}, Path(tmp_path / "m.pkl.gz"))[0]
    model.compute_metadata(copy.deepcopy(data))
    nn_ = model.build_neural_module()
    layers = [m for m in nn_.modules() if isinstance(m, TransformerEncoderLayer)]
    assert len(layers) == 3 and all(l.head_dim == 16 for l in layers)
    assert model.tensorize(copy.deepcopy(data[0])) is not None
    from buglab.models.layers.gru import BiGRULayer

    g = load_model({"modelName": "seq-gru", "hidden_state_size": 64, "num_layers": 2}, Path(tmp_path / "g.pkl.gz"))[0]
    g.compute_metadata(copy.deepcopy(data))
    gl = [m for m in g.build_neural_module().modules() if isinstance(m, BiGRULayer)]
    assert len(gl) == 2 and all(l.hidden_size == 32 and l.input_size == 64 for l in gl)
    with pytest.raises(AssertionError):
        TransformerEncoderLayer(65, 4)


def test_registry_passes_subtoken_combination_to_the_embedder(tmp_path):
    """node_representations["subtoken_combination"] (reference modelregistry.py:65-66: "max" unless given): max / sum / mean build,
    anything else is refused at construction."""
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models.layers.messagepassing import SubtokenEmbedder
    from buglab.models.modelregistry import load_model

    data = make_buglab_dataset(4, seed=2)
    for comb in ("max", "sum", "mean"):
        spec = {"modelName": "gnn-mlp", "hidden_state_size": 64, "node_representations": {"subtoken_combination": comb}}
        model = load_model(spec, Path(tmp_path / f"{comb}.pkl.gz"))[0]
        model.compute_metadata(copy.deepcopy(data))
        emb = [m for m in model.build_neural_module().modules() if isinstance(m, SubtokenEmbedder)]
        assert len(emb) == 1 and emb[0].subtoken_combination == comb
    with pytest.raises(ValueError):
        load_model({"modelName": "gnn-mlp", "node_representations": {"subtoken_combination": "median"}}, Path(tmp_path / "x.pkl.gz"))


def test_registry_passes_the_aggregation_function_to_the_layers(tmp_path):
    """`message_aggregation_function` in the model spec reaches every MlpMessagePassingLayer; combinations the HIP path does not run
    are refused at construction, not at the first step."""
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models.layers.messagepassing import MlpMessagePassingLayer
    from buglab.models.modelregistry import load_model

    data = make_buglab_dataset(4, seed=2)
    for agg in ("max", "sum", "mean"):
        model = load_model({"modelName": "gnn-mlp", "hidden_state_size": 64, "message_aggregation_function": agg}, Path(tmp_path / f"{agg}.pkl.gz"))[0]
        model.compute_metadata(copy.deepcopy(data))
        layers = [m for m in model.build_neural_module().modules() if isinstance(m, MlpMessagePassingLayer)]
        assert len(layers) == 8 and all(l.message_aggregation_function == agg for l in layers)
    bad = load_model({"modelName": "gnn-mlp", "hidden_state_size": 64, "message_aggregation_function": "sum",
                      "message_activation_placement": "message"}, Path(tmp_path / "bad.pkl.gz"))[0]
    bad.compute_metadata(copy.deepcopy(data))
    with pytest.raises(NotImplementedError):
        bad.build_neural_module()
    with pytest.raises(NotImplementedError):
        load_model({"modelName": "ggnn", "message_aggregation_function": "sum"}, Path(tmp_path / "g.pkl.gz"))
    with pytest.raises(ValueError):
        MlpMessagePassingLayer(64, 64, 64, 4, message_aggregation_function="median")


@pytest.mark.parametrize("name", ["seq-transformer", "seq-gru"])
def test_checkpoint_roundtrip_of_the_new_sequence_models(tmp_path, name):
    """save / restore_model (reference AbstractNeuralModel.save / restore_model as train.py:96 and evaluate.py use them): the pickled
    pair comes back with identical parameters and the same layer classes."""
    from buglab.data.synthetic import make_buglab_seq_dataset
    from buglab.models.modelregistry import load_model
    from buglab.models.seqmodel import SeqBugLabModel

    data = make_buglab_seq_dataset(4, seed=2)
    model = load_model({"modelName": name, "hidden_state_size": 64, "num_layers": 2, "num_heads": 4, "intermediate_dimension_size": 96},
                       Path(tmp_path / "m.pkl.gz"))[0]
    model.compute_metadata(copy.deepcopy(data))
    torch.manual_seed(3)
    nn_ = model.build_neural_module()
    model.save(tmp_path / "m.pkl.gz", nn_)
    model2, nn2 = SeqBugLabModel.restore_model(tmp_path / "m.pkl.gz", torch.device("cpu"))
    a, b = dict(nn_.named_parameters()), dict(nn2.named_parameters())
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    assert [type(m).__name__ for m in nn_.modules()] == [type(m).__name__ for m in nn2.modules()]
    assert model2.tensorize(copy.deepcopy(data[0])) is not None

"""CPU: the oracle's head / segment / loss restatement against vectors produced by the
reference's own code (tests/golden/make_golden.py).  Pins SURVEY section 8a rows H1-H8."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import buglab_oracle as O
from tests.refmap import MAP, _tx, golden_minibatch, head_params_from_golden


def test_scatter_log_softmax_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "heads_logsoftmax.npz"))
    out = O.scatter_log_softmax(torch.from_numpy(z["src"]), torch.from_numpy(z["index"]))
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_heads_forward_backward_match_reference(golden_dir, case):
    """a, b: default configuration (b with a buggy-sample weight); c: abstain_weight > 0
    (localizationmodule.py:95-100); d: use_all_gnn_layer_outputs -- the summarisation Linear over the
    concatenated layer states (gnn.py:68-74,118-121)."""
    z = np.load(os.path.join(golden_dir, f"heads_forward_{case}.npz"))
    params = {k: v.requires_grad_(True) for k, v in head_params_from_golden(z).items()}
    table = torch.from_numpy(z["node_states"]).requires_grad_(True)
    h = table
    if len(z["mp_dims"]):
        sW = torch.from_numpy(np.ascontiguousarray(z["w__GnnBugLabModule__summarization_layer.weight"].T)).requires_grad_(True)
        sb = torch.from_numpy(z["w__GnnBugLabModule__summarization_layer.bias"]).requires_grad_(True)
        h = table @ sW + sb
    mb = golden_minibatch(z)
    refs = mb["graph_data"]["reference_node_ids"]
    L = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int64)
    swap_lp, text_lp, var_lp, sel, _ = O.repair_logprobs(
        params, h, refs, mb["target_rewrites"], mb["rewrite_to_location_group"],
        mb["candidate_symbol_to_location_group"], mb["swapped_pair_to_call_location_group"])
    loc_loss, loc_lp, stats = O.localization_loss(
        params, h[L(refs["candidate_nodes"])], mb["graph_data"]["reference_node_graph_idx"]["candidate_nodes"],
        mb["has_bug"], mb["correct_candidate_node_idxs"], float(z["buggy_weight"]), float(z["abstain_weight"]))
    repair = -(text_lp[L(mb["correct_rewrite_idxs"])].sum() + var_lp[L(mb["correct_candidate_symbols"])].sum()
               + swap_lp[L(mb["correct_swapped_pair"])].sum()) * float(z["buggy_weight"])
    loss = loc_loss + repair / int(z["B"])
    np.testing.assert_allclose(loc_lp.detach().numpy(), z["loc_logprobs"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(text_lp.detach().numpy(), z["text_logprobs"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(var_lp.detach().numpy(), z["var_logprobs"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(swap_lp.detach().numpy(), z["swap_logprobs"], atol=2e-6, rtol=0)
    assert abs(float(loss) - float(z["loss"])) < 2e-6
    assert abs(stats["num_correct"] / int(z["B"]) - float(z["metrics_loc_accuracy"])) < 1e-9
    loss.backward()
    np.testing.assert_allclose(table.grad.numpy(), z["grad_node_states"], atol=2e-6, rtol=1e-5)
    if len(z["mp_dims"]):
        np.testing.assert_allclose(sW.grad.numpy(), z["g__GnnBugLabModule__summarization_layer.weight"].T, atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(sb.grad.numpy(), z["g__GnnBugLabModule__summarization_layer.bias"], atol=2e-6, rtol=1e-5)
    for ours, (ref, how) in MAP.items():
        np.testing.assert_allclose(params[ours].grad.numpy(), _tx(z["g_" + ref], how), atol=2e-6, rtol=1e-5, err_msg=ours)


def test_logprobs_normalise():
    """The reference's own sanity comments (basemodel.py:257-258, 342-344): per-graph
    localization probabilities sum to one."""
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples

    cfg = O.OracleConfig(hidden=32, num_layers=4, num_edge_types=4, vocab_size=100)
    mb = collate_samples(make_samples(3, seed=1, num_nodes=40, num_messages=150, num_edge_types=4, vocab_size=100, num_candidates=5), 4)
    out = O.forward_loss(O.init_params(cfg), mb, cfg)
    idx = np.concatenate([mb["graph_data"]["reference_node_graph_idx"]["candidate_nodes"], np.arange(3)])
    for b in range(3):
        assert abs(float(torch.logsumexp(out["loc_logprobs"][torch.from_numpy(idx == b)], 0))) < 1e-5


@pytest.mark.parametrize("loss_type", ["norm-kl", "norm-rmse", "classify-max-loss", "expectation"])
def test_generator_loss_matches_reference(golden_dir, loss_type):
    """Selector branch (reference gnn.py:189-219 + utils.py:101-179) -- SURVEY section 8f rank 2."""
    z = np.load(os.path.join(golden_dir, "heads_generator.npz"))
    params = head_params_from_golden(z)
    h = torch.from_numpy(z["node_states"]).requires_grad_(True)
    mb = golden_minibatch(z)
    cfg = O.OracleConfig(hidden=int(z["H"]))
    loss = O.generator_forward_loss(params, mb, cfg, loss_type, node_reprs=h)
    assert abs(float(loss.detach()) - float(z["loss_" + loss_type])) < 2e-6
    loss.backward()
    np.testing.assert_allclose(h.grad.numpy(), z["grad_node_states_" + loss_type], atol=2e-6, rtol=1e-4)


def test_message_activation_placement_hand_checked():
    """The one spec point the reference's call site leaves to ptgnn's defaults (gnnlayerdefs.py:6-23 passes no activation):
    GELU on every message before the max ("message") or on the aggregated [N, Dm] tensor ("aggregated", default).  A toy layer
    whose messages are known numbers: node 0 receives raw messages {-3.0, -0.7}, node 1 {0.5, 2.0}, node 2 nothing.
    gelu(-3.0) = -0.00405 > gelu(-0.7) = -0.1694: per-message placement routes node 0 to the -3.0 message, the aggregated
    placement to the -0.7 one -- value AND routing differ; on the monotone side (node 1) they agree."""
    import math

    h = torch.tensor([[-3.0], [-0.7], [0.5], [2.0]], dtype=torch.float64)  # 4 nodes, Din = 1
    W = torch.tensor([[[1.0], [0.0]]], dtype=torch.float64)  # one type: message = h[src]
    src, tgt = np.array([0, 1, 2, 3], np.int64), np.array([0, 0, 1, 1], np.int64)
    ident = dict(ln_g=torch.ones(1, dtype=torch.float64), ln_b=torch.zeros(1, dtype=torch.float64))
    gelu = lambda x: 0.5 * x * (1.0 + math.erf(x / math.sqrt(2.0)))
    for placement, want_agg, want_arg in (("aggregated", [gelu(-0.7), gelu(2.0), 0.0, 0.0], [1, 3, 4, 4]),
                                          ("message", [gelu(-3.0), gelu(2.0), 0.0, 0.0], [0, 3, 4, 4])):
        trace = []
        O.mp_layer(h, W, ident["ln_g"], ident["ln_b"], torch.ones(1, 1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64), src, tgt,
                   np.array([0, 4]), "gelu", 0.0, None, 1, trace=trace, msg_act_placement=placement)
        np.testing.assert_allclose(trace[0]["agg"].numpy().ravel(), want_agg, atol=1e-12)
        assert trace[0]["arg"].numpy().ravel().tolist() == want_arg  # 4 = E marks an empty segment
    # without an activation the placement is irrelevant
    outs = [O.mp_layer(h, W, ident["ln_g"], ident["ln_b"], torch.ones(1, 1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64), src, tgt,
                       np.array([0, 4]), "none", 0.0, None, 1, msg_act_placement=pl) for pl in ("aggregated", "message")]
    assert torch.equal(outs[0], outs[1])
    assert O.OracleConfig().msg_act_placement == "aggregated" and O.MpSpec(1, 1, 1).msg_act_placement == "aggregated"


def test_sum_and_mean_aggregation_hand_checked():
    """ptgnn's other message_aggregation_function values in the oracle: with identity-like weights the aggregate of a node is the plain
    sum (mean) of its incoming messages, a node without messages gets 0, and "max" is untouched by the new argument."""
    h = torch.tensor([[1.0, 2.0], [3.0, -4.0], [5.0, 6.0]])
    W = torch.zeros(1, 4, 2)
    W[0, 0, 0] = W[0, 1, 1] = 1.0  # message = h[src]
    src, tgt, tp = np.array([0, 1, 2]), np.array([2, 2, 0]), np.array([0, 3])
    ln_g, ln_b, Wd, bd = torch.ones(2), torch.zeros(2), torch.eye(2), torch.zeros(2)
    out = {}
    for agg in ("max", "sum", "mean"):
        tr = []
        O.mp_layer(h, W, ln_g, ln_b, Wd, bd, src, tgt, tp, "none", 0.0, None, 1, trace=tr, aggregation=agg)
        out[agg] = tr[0]["agg"].numpy()
    np.testing.assert_array_equal(out["max"], [[5.0, 6.0], [0.0, 0.0], [3.0, 2.0]])
    np.testing.assert_array_equal(out["sum"], [[5.0, 6.0], [0.0, 0.0], [4.0, -2.0]])
    np.testing.assert_array_equal(out["mean"], [[5.0, 6.0], [0.0, 0.0], [2.0, -1.0]])
    assert O.OracleConfig().msg_aggregation == "max"


def test_oracle_layers_equal_compositions_of_torch_modules():
    """What CAN be pinned of M0 - M4 without ptgnn: the oracle's layers against the torch.nn modules ptgnn's layers are assembled from
    (nn.Linear per edge type without bias, nn.GELU, nn.LayerNorm, nn.Linear, nn.Tanh for the MLP layer; nn.GRUCell for the gated one;
    nn.Embedding for the embedder), composed in the frozen spec's order, with torch.scatter_reduce as the aggregation: values and
    gradients agree to rounding.  This pins the ARITHMETIC of every piece to torch's own modules; the ORDER of the pieces is the
    written spec and stays unpinned."""
    torch.manual_seed(0)
    N, E, T, Din, Dm, Dout = 23, 90, 3, 8, 12, 8
    rng = np.random.default_rng(0)
    type_ptr = np.array([0, 40, 40, 90])  # (an empty type)
    src, tgt = rng.integers(0, N, E), rng.integers(0, N - 3, E)  # (the last three nodes receive nothing)
    h = torch.randn(N, Din, dtype=torch.float64, requires_grad=True)
    lin = [torch.nn.Linear(2 * Din, Dm, bias=False).double() for _ in range(T)]
    ln, dense = torch.nn.LayerNorm(Dm).double(), torch.nn.Linear(Dm, Dout).double()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.3, 0.3)
    for agg_name, reduce in (("max", "amax"), ("sum", "sum"), ("mean", "mean")):
        # --- torch.nn composition
        msgs = torch.cat([lin[t](torch.cat([h[src[type_ptr[t]:type_ptr[t + 1]]], h[tgt[type_ptr[t]:type_ptr[t + 1]]]], dim=-1)) for t in range(T)])
        idx = torch.as_tensor(tgt).view(E, 1).expand(E, Dm)
        agg = torch.zeros(N, Dm, dtype=torch.float64).scatter_reduce(0, idx, msgs, reduce=reduce, include_self=False)
        want = torch.tanh(dense(ln(torch.nn.functional.gelu(agg))))
        gw = torch.autograd.grad(want.square().sum(), [h] + [l.weight for l in lin] + [ln.weight, ln.bias, dense.weight, dense.bias])
        # --- oracle
        W = torch.stack([l.weight.t() for l in lin]).detach().requires_grad_(True)
        p = [t.detach().clone().requires_grad_(True) for t in (h, ln.weight, ln.bias, dense.weight.t(), dense.bias)]
        got = O.mp_layer(p[0], W, p[1], p[2], p[3], p[4], src, tgt, type_ptr, "gelu", 0.0, None, 1, aggregation=agg_name)
        go = torch.autograd.grad(got.square().sum(), [p[0], W, p[1], p[2], p[3], p[4]])
        assert float((got - want).abs().max()) < 1e-12, agg_name
        assert float((go[0] - gw[0]).abs().max()) < 1e-10
        for t in range(T):
            assert float((go[1][t] - gw[1 + t].t()).abs().max()) < 1e-10
        assert float((go[2] - gw[1 + T]).abs().max()) < 1e-10 and float((go[3] - gw[2 + T]).abs().max()) < 1e-10
        assert float((go[4] - gw[3 + T].t()).abs().max()) < 1e-10 and float((go[5] - gw[4 + T]).abs().max()) < 1e-10
    # --- gated layer: nn.GRUCell(input = aggregate, hidden = h)
    D = 8
    cell = torch.nn.GRUCell(Dm, D).double()
    Wg = torch.randn(T, D, Dm, dtype=torch.float64)
    hh = torch.randn(N, D, dtype=torch.float64)
    m = torch.cat([hh[src[type_ptr[t]:type_ptr[t + 1]]] @ Wg[t] for t in range(T)])
    agg = torch.zeros(N, Dm, dtype=torch.float64).scatter_reduce(0, torch.as_tensor(tgt).view(E, 1).expand(E, Dm), m, reduce="amax", include_self=False)
    want = cell(agg, hh)
    got = O.gated_mp_layer(hh, Wg, cell.weight_ih.t(), cell.bias_ih, cell.weight_hh.t(), cell.bias_hh, src, tgt, type_ptr, 0.0, None, 1)
    assert float((got - want).abs().max()) < 1e-12
    # --- embedder: nn.Embedding rows, max over the real subtokens
    emb = torch.nn.Embedding(50, 6).double()
    ids, lens = rng.integers(0, 50, (N, 4)), rng.integers(1, 5, N)
    e = emb(torch.as_tensor(ids)).masked_fill(torch.arange(4)[None, :, None] >= torch.as_tensor(lens)[:, None, None], -math.inf)
    assert torch.equal(O.embed_nodes(emb.weight, ids, lens, 0.0, None), e.max(dim=1).values)


def test_embedder_dropout_placement_hand_checked():
    """Dropout of the subtoken embedder before or after the max over subtokens (oracle.embed_nodes): with every subtoken of a node
    kept the two placements give the same values; a dropped winner lets another subtoken (or the 0 of a dropped one) win only
    under "before_pooling"; eval mode (no seed) does not depend on the placement."""
    table = torch.tensor([[0.0, 0.0], [1.0, -1.0], [2.0, -3.0], [-5.0, 4.0]])
    ids, lens = np.array([[1, 2, 3], [3, 0, 0]]), np.array([3, 1])
    for pl in ("after_pooling", "before_pooling"):
        np.testing.assert_array_equal(O.embed_nodes(table, ids, lens, 0.5, None, pl).numpy(), [[2.0, 4.0], [-5.0, 4.0]])
    seed, p = 3, 0.5
    keep_rows = O.dropout_keep_mask(seed, 0, 4, p).reshape(2, 2)
    keep_subs = O.dropout_keep_mask(seed, 0, 12, p).reshape(2, 3, 2)
    after = O.embed_nodes(table, ids, lens, p, seed, "after_pooling").numpy()
    np.testing.assert_allclose(after, np.array([[2.0, 4.0], [-5.0, 4.0]]) * keep_rows / (1 - p))
    emb = table.numpy()[ids] * keep_subs / (1 - p)
    emb[1, 1:] = -np.inf  # padding slots of the one-subtoken node
    np.testing.assert_allclose(O.embed_nodes(table, ids, lens, p, seed, "before_pooling").numpy(), emb.max(axis=1))
    assert O.OracleConfig().embed_dropout_placement == "after_pooling"
    # the other subtoken combinations (node_representations["subtoken_combination"]): sums over the REAL subtokens only
    np.testing.assert_array_equal(O.embed_nodes(table, ids, lens, 0.5, None, "after_pooling", "sum").numpy(), [[-2.0, 0.0], [-5.0, 4.0]])
    np.testing.assert_allclose(O.embed_nodes(table, ids, lens, 0.5, None, "after_pooling", "mean").numpy(), [[-2.0 / 3, 0.0], [-5.0, 4.0]])
    np.testing.assert_allclose(O.embed_nodes(table, ids, lens, p, seed, "before_pooling", "sum").numpy(),
                               np.where(np.arange(3)[None, :, None] < lens[:, None, None], table.numpy()[ids] * keep_subs / (1 - p), 0.0).sum(axis=1))


def test_config_c1_plumbing_on_cpu_oracle():
    """BASELINE.json configs[0]: gnn-mlp, hidden 128, 4 layers, ONE graph of ~500 nodes, CPU.  The product has no CPU path
    (it fails loudly without the GPU), so the CPU-runnable case is the oracle's: a few clip + Adam steps on that batch
    lower the loss, log-probabilities stay normalised."""
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples

    cfg = O.OracleConfig(hidden=128, num_layers=4, num_edge_types=16, vocab_size=2000)
    mb = collate_samples(make_samples(1, seed=21, num_nodes=500, num_messages=2500, num_edge_types=16, vocab_size=2000, buggy=True), 16)
    params = O.init_params(cfg, seed=0)
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v) for k, v in params.items()}
    losses = []
    for step in range(1, 5):
        out, grads = O.forward_backward(params, mb, cfg)
        losses.append(float(out["loss"]))
        O.adam_clip_step(params, grads, m, v, step, lr=1e-2, warmup=0)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    lp = out["loc_logprobs"]
    assert abs(float(torch.logsumexp(lp, 0))) < 1e-5  # one graph: candidates + NO_BUG form one distribution


def test_optimiser_trajectory_matches_the_reference_pieces(golden_dir):
    """T1: oracle.adam_clip_step against a trajectory produced by the reference's own `optimizer()` +
    `LinearWarmupScheduler` (utils.py:51-66) + torch's Adam and clip_grad_norm_(0.5), stepped optimiser-then-scheduler
    like ptgnn's trainer (tests/golden/make_golden_optim.py).  The HIP optimiser is compared with this oracle function on
    the GPU (tests/test_hip_kernels.py::test_flat_adam_matches_oracle), which closes the chain to the reference."""
    z = np.load(os.path.join(golden_dir, "optim_trajectory.npz"))
    params = {0: torch.from_numpy(z["init0"].copy()), 1: torch.from_numpy(z["init1"].copy())}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    for k in range(int(z["steps"])):
        grads = {0: torch.from_numpy(z["grads0"][k].copy()), 1: torch.from_numpy(z["grads1"][k].copy())}
        O.adam_clip_step(params, grads, m, v, k + 1, lr=float(z["lr"]), clip=float(z["clip"]), warmup=int(z["warmup"]))
        for i in (0, 1):
            want = torch.from_numpy(z[f"after{i}"][k])
            assert (params[i] - want).abs().max() < 2e-6, (k, i, float((params[i] - want).abs().max()))
    # the first step ran with learning-rate factor 0 (LambdaLR semantics): parameters unchanged after it
    assert np.array_equal(z["after0"][0], z["init0"])

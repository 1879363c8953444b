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

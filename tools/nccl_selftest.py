#!/usr/bin/env python
"""World-size-1 `nccl` (= RCCL) self-test of the data-parallel code path: init_process_group over 127.0.0.1, the trainer's
`broadcast_parameters`, one `FlatAdam.step_data_parallel` (the ONE all-reduce carrying gradient, graph count and has-batch
flag, bucketed form included) against the plain single-process `step` on the same gradients, an idle step, teardown.
An 8-GPU node only ever sees this code through the driver's scaling run; this makes sure the RCCL calls themselves have
executed once on the GPU box before that.  Prints NCCL_SELFTEST_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1"})
os.environ.setdefault("MASTER_PORT", "29731")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from buglab.data.collate import collate_samples, to_device
from buglab.data.synthetic import make_samples
from buglab.models import hip_ops
from buglab.models.gnn import build_gnn_mlp_module
from buglab.runtime.optim import FlatAdam


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    hip_ops.set_deterministic(True)  # fixed summation orders: the two trajectories below then differ by the rounding of (B g) / B only
    try:
        mb = to_device(collate_samples(make_samples(4, seed=3, num_nodes=120, num_messages=600, num_edge_types=5, vocab_size=300), 5), "cuda")
        results = {}
        for mode in ("single", "dp"):
            torch.manual_seed(0)
            module = build_gnn_mlp_module(64, 4, 5, vocabulary_size=300, dropout_rate=0.0).cuda().train()
            opt = FlatAdam(module.parameters(), num_warmup_steps=0)
            opt.distributed = True
            if mode == "dp":
                opt.broadcast_parameters(0)
                # the layer-wise bucketed reduction (asynchronous all-reduces on a communication stream, issued from the
                # backward notifications) only arms itself for world sizes > 1: force it, so that this path runs on RCCL too
                opt._dp_active = lambda: True
                assert opt.set_overlap_groups(module.overlap_parameter_groups())
            for _ in range(3):
                opt.zero_grad()
                if mode == "dp" and hasattr(opt, "begin_data_parallel_step"):
                    opt.begin_data_parallel_step(4)
                loss = module(**mb, dropout_seed=1)
                loss.backward()
                if mode == "dp":
                    opt.step_data_parallel(4)
                else:
                    opt.step()
            if mode == "dp":  # an idle step (no rank had a minibatch) leaves the parameters untouched and is detected
                before = opt.flat_param.clone()
                opt.zero_grad()
                if hasattr(opt, "begin_data_parallel_step"):
                    opt.begin_data_parallel_step(0)
                opt.step_data_parallel(0)
                assert opt.previous_step_was_idle()
                assert torch.equal(before, opt.flat_param)
            torch.cuda.synchronize()
            results[mode] = opt.flat_param.clone()
        # world size 1: sum over ranks of B * g / B == g up to the rounding of (B * g) / B -- equal to ~1 ulp of the update
        diff = float((results["single"] - results["dp"]).abs().max())
        assert diff < 5e-6, diff
        t = torch.ones(1 << 20, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        torch.cuda.synchronize()
        assert float(t.sum()) == float(1 << 20)
        assert opt._buckets and opt._comm_stream is not None, "the bucketed path did not run"
        print(f"NCCL_SELFTEST_OK backend={dist.get_backend()} buckets={len(opt._buckets)} max |param diff| single vs data-parallel step: {diff:.2e}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Data-parallel path on a device: W ranks (torchrun; W = 2 ... 8) train gnn-mlp on unequal shards of one minibatch and must
end up exactly where ONE process training on the union minibatch ends up.

    BL_FORCE_DEVICE=0 BL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29533 tools/dp_check.py
    python tools/dp_check.py --ranks 8        # the same: launches itself like that

On a single-GPU box all ranks share GPU 0 and the collective runs over gloo (RCCL refuses two ranks on one device);
the code path is the product's: FlatAdam.step_data_parallel = ONE all-reduce of [B_rank x gradient | B_rank, flag] per
step, fused clip + Adam reading the global count on the device.

Two ranks run out of data at DIFFERENT steps (the last rank one step before the end, the one before it -- when W > 2 --
two steps before): a rank without a minibatch never runs backward and still issues every bucket's collective in order.
Rank 0 also trains a single process in lockstep on the union of the shards that still have data and compares, every step: the graph-weighted loss, the reduced gradient / global graph count, and the
parameters after the update.  Before each step the single process is given the replicas' parameters and moments: the
routed max of the message-passing layers is discontinuous, so 1e-7 of summation-order difference in the parameters
flips a near-tie now and then and moves a gradient entry by 1e-3 -- a property of the model (tests/test_hip_parity.py
handles it the same way), not of the reduction that is being checked here.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd")]
import torch
import torch.distributed as dist

from buglab.data.collate import collate_samples, to_device
from buglab.data.synthetic import make_samples
from buglab.models import hip_ops
from buglab.models.gnn import build_gnn_mlp_module
from buglab.runtime import distributed as D
from buglab.runtime.optim import FlatAdam


STEPS = 5


def shard_sizes(world: int):
    """unequal numbers of graphs per rank: (5, 3) at two ranks like rounds 2-5, then 2 ... 6"""
    return tuple((5, 3, 4, 2, 6, 3, 4, 2)[r % 8] for r in range(world))


def has_data(rank: int, world: int, step: int) -> bool:
    """the last rank runs dry at the last step, the one before it (if it is not rank 0) already a step earlier"""
    if rank == world - 1 and step >= STEPS - 1:
        return False
    if world > 2 and rank == world - 2 and step >= STEPS - 2:
        return False
    return True


def self_launch(nproc: int) -> int:
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("BL_FORCE_DEVICE", "0")
    env.setdefault("BL_DIST_BACKEND", "gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)], env=env)


def main():
    if "--ranks" in sys.argv and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(int(sys.argv[sys.argv.index("--ranks") + 1])))
    rank, world, device = D.init_from_env("cuda")
    assert 2 <= world <= 8
    hip_ops.load_library()
    hip_ops.use_step_stream(device)  # what ModelTrainer.train does: the step's chain on the high-priority stream
    sizes = shard_sizes(world)
    samples = make_samples(sum(sizes), seed=7, num_nodes=300, num_messages=1500, num_edge_types=8, vocab_size=2000)
    lo = sum(sizes[:rank])
    mine = to_device(collate_samples(samples[lo:lo + sizes[rank]], 8), device)
    build = lambda: build_gnn_mlp_module(64, 8, 8, vocabulary_size=2000, dropout_rate=0.0).to(device).train()
    torch.manual_seed(1234 + rank)  # replicas start DIFFERENT; broadcast_parameters must fix that
    module = build()
    opt = FlatAdam(module.parameters(), lr=1e-3, num_warmup_steps=0)
    opt.broadcast_parameters(0)
    buckets = os.environ.get("DP_CHECK_BUCKETS", "1") != "0"
    if buckets:  # the trainer's layer-wise buckets: reduced asynchronously behind the backward pass, fixed order on every rank
        assert opt.set_overlap_groups(module.overlap_parameter_groups())

    def replica_gap(t):
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t.detach().clone())
        return max(float((every[0] - other).abs().max()) for other in every[1:])

    assert replica_gap(opt.flat_param) == 0.0, "broadcast_parameters left the replicas different"
    # rank 0 also trains ONE process' worth in lockstep: the union of the shards that have data at that step
    if rank == 0:
        torch.manual_seed(1234)
        ref = build()
        ropt = FlatAdam(ref.parameters(), lr=1e-3, num_warmup_steps=0, distributed=False)
        unions = {}

        def union_at(step):
            live = tuple(r for r in range(world) if has_data(r, world, step))
            if live not in unions:
                picked = [smp for r in live for smp in samples[sum(sizes[:r]):sum(sizes[:r]) + sizes[r]]]
                unions[live] = to_device(collate_samples(picked, 8), device)
            return unions[live]

        names = [n for n, _ in ref.named_parameters()]
    report = []
    for step in range(STEPS):
        if rank == 0:  # the single process starts every step from the replicas' state (see the docstring)
            for mine_t, theirs_t in ((ropt.flat_param, opt.flat_param), (ropt.m, opt.m), (ropt.v, opt.v)):
                mine_t.copy_(theirs_t)
            ropt.step_count = opt.step_count
            hip_ops.invalidate_weight_packs()
        opt.zero_grad()
        live_now = has_data(rank, world, step)
        B, my_loss = (sizes[rank] if live_now else 0), 0.0
        if buckets:
            opt.begin_data_parallel_step(B)
        if live_now:
            loss = module(**mine)
            loss.backward()
            my_loss = float(loss.detach())
        opt.step_data_parallel(B)
        gaps = (replica_gap(opt.flat_grad), replica_gap(opt.sqnorm), replica_gap(opt.flat_param))
        assert gaps == (0.0, 0.0, 0.0), f"step {step}: replicas differ in (reduced gradient, its norm, parameters) by {gaps}"
        assert not (step > 0 and opt.previous_step_was_idle())
        weighted = torch.tensor([my_loss * B, float(B)], dtype=torch.float64)
        parts = [torch.zeros(2, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(parts, weighted.to(device))
        if rank == 0:
            dp_loss = float(sum(p[0] for p in parts) / sum(p[1] for p in parts))
            ropt.zero_grad()
            l = ref(**union_at(step))
            l.backward()
            hip_ops.join_side_stream()
            total = float(sum(p[1] for p in parts))
            g_gap = float((ropt.flat_grad - opt.flat_grad / total).abs().max())
            g_scale = float(ropt.flat_grad.abs().max())
            # the update is compared on the SAME input: the single process steps on the replicas' mean gradient.  (Adam's first
            # steps move a coordinate by lr * g / (|g| + 1e-8): where |g| ~ 1e-7 the 1e-7 of summation-order noise between the two
            # gradients -- checked just above -- would show as a tenth of lr; that is Adam, not the reduction or the count.)
            ropt.flat_grad.copy_(opt.flat_grad / total)
            ropt.step()
            torch.cuda.synchronize()
            gap_per_param = [float((a.detach() - b.detach()).abs().max()) for a, b in zip(ref.parameters(), module.parameters())]
            worst = max(range(len(names)), key=lambda i: gap_per_param[i])
            report.append((step, dp_loss, float(l.detach()), g_gap, g_scale, gap_per_param[worst], names[worst]))
    opt.zero_grad()
    if buckets:
        opt.begin_data_parallel_step(0)
    opt.step_data_parallel(0)  # nobody has data: the idle step that ends an epoch
    assert opt.previous_step_was_idle() and opt.step_count == STEPS
    torch.cuda.synchronize()
    assert replica_gap(opt.flat_param) == 0.0
    if rank == 0:
        dry = {r: min(st for st in range(STEPS + 1) if st == STEPS or not has_data(r, world, st)) for r in range(world)}
        print(f"dp_check: {world} ranks x {sizes} graphs (ranks running dry from step: { {r: st for r, st in dry.items() if st < STEPS} }) on {torch.cuda.get_device_name(0)} over {dist.get_backend()}, "
              f"{'layer-wise buckets (' + str(len(opt._buckets)) + ') behind backward' if buckets else 'one all-reduce per step'}; replicas bit-identical after "
              f"every step; against one process on the union minibatch:", flush=True)
        for step, dp_loss, ref_loss, g_gap, g_scale, p_gap, p_name in report:
            print(f"  step {step}: loss {dp_loss:.6f} (graph-weighted over ranks) vs {ref_loss:.6f}; max |gradient gap| {g_gap:.2e} "
                  f"(largest entry {g_scale:.2e}); max |parameter gap| after the step {p_gap:.2e} ({p_name})", flush=True)
        assert all(abs(r[1] - r[2]) < 2e-5 for r in report), "losses differ"
        assert all(r[3] < 2e-6 * max(1.0, r[4]) for r in report), "gradients differ"
        # same gradient in, same parameters out (global graph count, clip by the mean gradient's norm, Adam): a wrong count or a
        # lost shard would move every coordinate by ~lr = 1e-3
        assert all(r[5] < 5e-6 for r in report), "parameters differ"
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

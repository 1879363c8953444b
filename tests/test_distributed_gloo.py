"""CPU, world_size 2 over gloo: the data-parallel reduction used on N GPUs (one all-reduce of the
flat gradient buffer, per-rank weights B_rank / B_total) reproduces the single-process full-batch
gradient.  The per-rank gradient comes from the CPU oracle (tests may use it); the reduction code is
the product's (buglab.runtime.optim.FlatAdam.reduce_gradients, buglab.runtime.distributed)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import buglab_oracle as O
from tests import helpers as Hh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, args, nprocs=2, timeout=150.0):
    """mp.spawn with a deadline: a rank stuck in a collective fails the test instead of hanging the suite."""
    import time

    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.monotonic() + timeout
    try:
        while not ctx.join(timeout=1.0):
            if time.monotonic() > deadline:
                raise TimeoutError(f"{fn.__name__}: ranks still running after {timeout:.0f} s")
    finally:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()


def _worker(rank, world, port, sizes, out_path):
    import sys

    from tests.conftest import PKG, ROOT  # noqa: F401 (sys.path side effect)
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples
    from buglab.runtime import distributed as D
    from buglab.runtime.optim import FlatAdam

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, dev = D.init_from_env("cpu")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.set_num_threads(2)
    cfg = O.OracleConfig(hidden=32, num_layers=4, num_edge_types=4, vocab_size=120)
    samples = make_samples(sum(sizes), seed=11, num_nodes=40, num_messages=160, num_edge_types=4, vocab_size=120, num_candidates=6)
    lo = sum(sizes[:rank])
    mine = samples[lo:lo + sizes[rank]]
    params = O.init_params(cfg, seed=0)
    _, grads = O.forward_backward(params, collate_samples(mine, 4), cfg)
    ps = [torch.nn.Parameter(v.clone()) for v in params.values()]
    opt = FlatAdam(ps)
    for p, g in zip(ps, grads.values()):
        p.grad.copy_(g)
    weight = D.global_batch_weight(len(mine), dev)
    assert abs(weight - sizes[rank] / sum(sizes)) < 1e-12
    rest = opt.reduce_gradients(weight)
    assert rest == 1.0
    assert D.max_over_ranks(float(rank), dev) == world - 1
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in zip(params, ps)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(3, 3), (4, 2)])
def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path, sizes):
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples

    out = str(tmp_path / "g.pt")
    _spawn(_worker, (2, _free_port(), sizes, out))
    reduced = torch.load(out)
    cfg = O.OracleConfig(hidden=32, num_layers=4, num_edge_types=4, vocab_size=120)
    samples = make_samples(sum(sizes), seed=11, num_nodes=40, num_messages=160, num_edge_types=4, vocab_size=120, num_candidates=6)
    _, full = O.forward_backward(O.init_params(cfg, seed=0), collate_samples(samples, 4), cfg)
    for k in full:
        assert Hh.maxdiff(reduced[k], full[k]) <= 1e-5 * float(full[k].abs().max()) + 1e-7, k


def test_partition_by_messages_balances_load():
    from buglab.runtime.distributed import partition_by_messages

    msgs = [10, 1000, 30, 990, 500, 480, 20, 5]
    parts = partition_by_messages(msgs, 2)
    assert sorted(i for p in parts for i in p) == list(range(8))
    loads = [sum(msgs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 40


def test_partition_by_messages_at_world_size_8():
    """8 ranks, graphs dealt by message count.  BASELINE configs[2] / [3] (256 graphs of 10 000 messages
    each, uniform or power-law in-degree) split exactly; a PyPI-like population of very unequal graphs (log-normal sizes over
    two decades) stays within 3 % between the heaviest and the lightest rank -- the step time follows the messages."""
    from buglab.data.synthetic import make_samples
    from buglab.runtime.distributed import partition_by_messages

    c4 = make_samples(32, seed=3, num_nodes=500, num_messages=2500, num_edge_types=16, degree="powerlaw", max_degree=512)
    msgs = [int(sum(len(a) for a in s.graph_data.adjacency_lists)) for s in c4]
    parts = partition_by_messages(msgs, 8)
    assert sorted(i for p in parts for i in p) == list(range(32)) and {len(p) for p in parts} == {4}
    loads = [sum(msgs[i] for i in p) for p in parts]
    assert max(loads) == min(loads)
    rng = np.random.default_rng(1)
    msgs = np.exp(rng.normal(np.log(8000.0), 1.0, 256)).astype(np.int64).clip(200, 200_000).tolist()
    parts = partition_by_messages(msgs, 8)
    assert sorted(i for p in parts for i in p) == list(range(256))
    loads = [sum(msgs[i] for i in p) for p in parts]
    assert (max(loads) - min(loads)) / max(loads) <= 0.03, loads
    by_count = [sum(msgs[i::8]) for i in range(8)]  # what dealing by graph COUNT would give
    assert (max(by_count) - min(by_count)) / max(by_count) > 0.15


def test_balanced_rank_share_is_a_partition_by_messages():
    from buglab.runtime.distributed import balanced_rank_share

    rng = np.random.default_rng(0)
    data = [{"id": i, "graph": {"edges": {"a": [0] * int(rng.integers(1, 400)), "b": [0] * int(rng.integers(1, 50))}}} for i in range(203)]
    shares = [list(balanced_rank_share(iter(data), r, 4)) for r in range(4)]
    assert sorted(d["id"] for s in shares for d in s) == list(range(203))  # every datapoint exactly once
    load = [sum(len(d["graph"]["edges"]["a"]) + len(d["graph"]["edges"]["b"]) for d in s) for s in shares]
    assert max(load) - min(load) < 0.06 * max(load)  # i % world would leave this at ~15 %
    assert list(balanced_rank_share(iter(data), 0, 1)) == data


# ---- the training loop itself under data parallelism (ranks with UNEQUAL numbers of minibatches) ------------------------------
class _TinyModel:
    """Host-side stand-in with the AbstractNeuralModel surface ModelTrainer uses; a minibatch = a [B, 3] feature block."""

    def __init__(self, per_rank_batches):
        self.per_rank_batches = per_rank_batches

    def tensorize_dataset(self, data, parallelize=False):
        return iter(data)

    def minibatch_iterator(self, tensors, device, size, parallelize=False):
        for x in tensors:
            yield {"x": torch.tensor(x, dtype=torch.float32), "has_bug": torch.zeros(len(x), dtype=torch.bool)}, None

    def save(self, path, nn):
        torch.save({k: v.clone() for k, v in nn.state_dict().items()}, path)


class _TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(3))
        self.b = torch.nn.Parameter(torch.zeros(1))

    def forward(self, *, x, has_bug):
        return ((x @ self.w + self.b - x.sum(1)) ** 2).mean()

    def reset_metrics(self):
        pass

    def report_metrics(self):
        return {}


def _cpu_flat_adam():
    from buglab.runtime.optim import FlatAdam

    class CpuFlatAdam(FlatAdam):
        """The product's data-parallel protocol with the two device kernels replaced by the same arithmetic in torch."""

        def _apply_update_data_parallel(self):
            bt = float(self.tail[0])
            if not bt > 0:
                return
            g = self.flat_grad / bt
            g = g * min(1.0, self.clip / (float(g.norm()) + 1e-6))
            self.m.mul_(self.beta1).add_(g, alpha=1 - self.beta1)
            self.v.mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
            bc1, bc2 = 1 - self.beta1 ** self.step_count, 1 - self.beta2 ** self.step_count
            self.flat_param.addcdiv_(self.m, (self.v.sqrt() / bc2 ** 0.5).add_(self.eps), value=-self.lr_at(self.step_count) / bc1)

    return CpuFlatAdam


def _trainer_worker(rank, world, port, batches, out_dir):
    from tests.conftest import PKG, ROOT  # noqa: F401
    from buglab.runtime import distributed as D
    from buglab.runtime.trainer import ModelTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_from_env("cpu")
    torch.manual_seed(100 + rank)  # deliberately different: the trainer must broadcast rank 0's parameters
    net = _TinyNet()
    with torch.no_grad():
        net.w.normal_()
    model = _TinyModel(batches)
    Opt = _cpu_flat_adam()
    opts = []

    def make_opt(params):
        opts.append(Opt(params, lr=0.05, num_warmup_steps=0))
        return opts[-1]

    tr = ModelTrainer(model, os.path.join(out_dir, f"m{rank}.pt"), max_num_epochs=2, optimizer_creator=make_opt)
    tr.neural_module = net
    tr._rank_share = lambda data: iter(data)  # each rank is handed its own (unequal) share below
    tr.train(batches[rank], batches[rank][:1], initialize_metadata=False, parallelize=False, device=torch.device("cpu"))
    torch.save({"w": net.w.detach().clone(), "b": net.b.detach().clone(), "steps": opts[0].step_count}, os.path.join(out_dir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_two_ranks_unequal_shards_stay_identical(tmp_path):
    """ModelTrainer.train over gloo, 2 ranks, rank 0 with 2 minibatches per epoch and rank 1 with 4: both ranks finish every
    epoch together (no hang), take the same number of optimiser steps, end with IDENTICAL parameters, and those equal a
    single-process run over the per-step unions of the two ranks' minibatches."""
    rng = np.random.default_rng(3)
    mk = lambda n: rng.normal(size=(n, 3)).tolist()
    batches = [[mk(4), mk(2)], [mk(3), mk(5), mk(2), mk(6)]]
    _spawn(_trainer_worker, (2, _free_port(), batches, str(tmp_path)))
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0["w"], p1["w"]) and torch.equal(p0["b"], p1["b"])
    assert p0["steps"] == p1["steps"] == 2 * 4  # 4 real steps per epoch; the idle step that ends an epoch is not counted
    # single process: step k of an epoch uses the concatenation of what the ranks had at step k
    torch.manual_seed(100)
    net = _TinyNet()
    with torch.no_grad():
        net.w.normal_()
    opt = _cpu_flat_adam()(net.parameters(), lr=0.05, num_warmup_steps=0)
    for _ in range(2):
        for k in range(4):
            xs = [torch.tensor(b[k], dtype=torch.float32) for b in batches if k < len(b)]
            opt.zero_grad()
            # per-rank mean losses weighted by B_rank / B_total == what the all-reduce of B_rank-scaled gradients computes
            tot = sum(len(x) for x in xs)
            loss = sum(net(x=x, has_bug=None) * (len(x) / tot) for x in xs)
            loss.backward()
            opt.tail[0] = 1.0
            opt.step_count += 1
            opt._apply_update_data_parallel()
    assert float((net.w.detach() - p0["w"]).abs().max()) < 1e-6 and float((net.b.detach() - p0["b"]).abs().max()) < 1e-6


# ---- layer-wise gradient buckets (FlatAdam.set_overlap_groups): same collectives, same order, on every rank ----------------
def _bucket_worker(rank, world, port, out_dir):
    from tests.conftest import PKG, ROOT  # noqa: F401
    from buglab.runtime import distributed as D

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_from_env("cpu")
    torch.set_num_threads(1)
    Opt = _cpu_flat_adam()
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 3), (5,), (4, 6), (6,), (3, 3), (2,), (9,)]  # "embedding", layer 1 (2 tensors), layer 2 (2), "heads" (2)
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    results = {}
    for mode in ("plain", "buckets"):
        opt = Opt([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=0.01, num_warmup_steps=0)
        if mode == "buckets":
            assert opt.set_overlap_groups([opt.params[1:3], opt.params[3:5]])
        # graphs per rank and step -- two ranks: (3, 2), (4, 0), (0, 0), (1, 5): at step 1 the odd ranks have no minibatch,
        # step 2 is idle everywhere; at eight ranks ranks 6 and 7 are also dry at step 3 (they ran out at different steps)
        plan = [lambda r: 3 - r % 2, lambda r: 0 if r % 2 else 4 - r // 2, lambda r: 0,
                lambda r: 0 if (world > 2 and r >= world - 2) else 1 + (4 * r + 4) % 7]
        for step, graphs_of in enumerate(plan):
            B = graphs_of(rank)
            opt.zero_grad()
            if mode == "buckets":
                opt.begin_data_parallel_step(B)
            if B > 0:
                gg = torch.Generator().manual_seed(100 * step + rank)
                for p in opt.params:
                    p.grad.copy_(torch.randn(p.shape, generator=gg))
                if mode == "buckets":
                    # backward reaches layer 2 first, then layer 1; on odd steps the notifications arrive tensor by tensor
                    from buglab.models import hip_ops

                    assert hip_ops.GRAD_READY_CALLBACK is not None
                    if step % 2:
                        for p in (opt.params[4], opt.params[1], opt.params[3], opt.params[2]):
                            hip_ops._notify_backward_launched((p,))
                    else:
                        hip_ops._notify_backward_launched(tuple(opt.params[3:5]))
                        hip_ops._notify_backward_launched(tuple(opt.params[1:3]))
            opt.step_data_parallel(B)
        results[mode] = (opt.flat_param.clone(), opt.flat_grad.clone(), opt.tail.clone(), opt.step_count)
        if mode == "buckets":
            # rank 0's backward "raises" after one layer has reported: abort_data_parallel_step disarms the callback and
            # completes the step's plan, so rank 1 (which steps normally) is not left hanging; the next step runs as usual
            from buglab.models import hip_ops

            opt.zero_grad()
            opt.begin_data_parallel_step(2)
            hip_ops._notify_backward_launched(tuple(opt.params[3:5]))
            if rank == 0:
                opt.abort_data_parallel_step()
                assert hip_ops.GRAD_READY_CALLBACK is None and opt._armed_B is None and opt._works == []
                hip_ops._notify_backward_launched(tuple(opt.params[1:3]))  # a later backward: no collective is issued
            else:
                hip_ops._notify_backward_launched(tuple(opt.params[1:3]))
                opt.step_data_parallel(2)
            opt.zero_grad()
            opt.begin_data_parallel_step(1)
            opt.step_data_parallel(1)
    if world == 2:  # one addition per element, whatever the collective's chunking: the two plans agree bit for bit
        assert torch.equal(results["plain"][0], results["buckets"][0]), "bucketed reduction must give bit-identical parameters"
        assert torch.equal(results["plain"][1], results["buckets"][1]) and torch.equal(results["plain"][2], results["buckets"][2])
    else:
        # more than two ranks: a ring / halving all-reduce adds the ranks' contributions of an element in an order that depends
        # on the element's position in the reduced buffer, so one long buffer and the per-layer buckets round differently
        # (1 ulp).  What must hold bit for bit is that every REPLICA sees the same sums (checked by the parent on the saved
        # parameters and here on the gradient); the two plans agree to rounding.
        every = [torch.zeros_like(results["buckets"][1]) for _ in range(world)]
        dist.all_gather(every, results["buckets"][1])
        assert all(torch.equal(every[0], e) for e in every[1:]), "replicas must hold identical reduced gradients"
        for a, b in zip(results["plain"][:3], results["buckets"][:3]):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
    assert results["plain"][3] == results["buckets"][3]
    torch.save(results["buckets"][0], os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_layerwise_gradient_buckets_equal_single_allreduce(tmp_path, world):
    """FlatAdam with overlap groups (one all-reduce per layer group issued from the backward notifications, last layer
    first, then the remaining ranges with the tail) against the single all-reduce: bit-identical parameters on every rank
    (2 and 8 ranks over gloo), including steps where some ranks have no minibatch (they issue the same collectives from
    step_data_parallel) and an idle step; no hang under out-of-order notifications or an aborted backward."""
    _spawn(_bucket_worker, (world, _free_port(), str(tmp_path)), nprocs=world, timeout=300.0)
    first = torch.load(tmp_path / "b0.pt")
    for r in range(1, world):
        assert torch.equal(first, torch.load(tmp_path / f"b{r}.pt")), r


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_itself_for_several_ranks(world):
    """`python bench.py --gpus N` with no launcher around it must become the launcher (torch.distributed.run on 127.0.0.1)
    and get both ranks through init_process_group -- the first hardware SCALE run may be started exactly like that.
    --dry-launch stops after one all-reduce (gloo here: no GPU)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-launch"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec == {"dry_launch": True, "ok": True, "n_gpus": world, "backend": "gloo", "rank_sum": world * (world - 1) / 2.0}

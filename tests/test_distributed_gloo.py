"""CPU, world_size 2 over gloo: the data-parallel reduction used on N GPUs (one all-reduce of the
flat gradient buffer, per-rank weights B_rank / B_total) reproduces the single-process full-batch
gradient.  The per-rank gradient comes from the CPU oracle (tests may use it); the reduction code is
the product's (buglab.runtime.optim.FlatAdam.reduce_gradients, buglab.runtime.distributed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import buglab_oracle as O
from tests import helpers as Hh


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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
    mp.spawn(_worker, args=(2, _free_port(), sizes, out), nprocs=2, join=True)
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

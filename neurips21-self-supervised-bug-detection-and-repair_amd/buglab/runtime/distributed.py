"""Graph-level data parallelism: one process per GPU, each rank collates
its own graphs, ONE RCCL all-reduce (sum) of the flat fp32 gradient buffer per step over xGMI --
`torch.distributed` backend "nccl" is RCCL on ROCm.  The reference has no multi-GPU code at all
(single `cuda:0`, modelregistry.py:155)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(device_type: str = "cuda") -> Tuple[int, int, torch.device]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun).  Returns
    (rank, world_size, device).  A single process without those variables is world_size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        if os.environ.get("BL_FORCE_DEVICE") is not None:  # testing only: several ranks on one GPU (needs gloo)
            local = int(os.environ["BL_FORCE_DEVICE"])
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BL_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, device


def partition_by_messages(num_messages: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy bin-packing of graphs onto ranks by message count (step time is proportional to E,
    not to the number of graphs)."""
    order = np.argsort(-np.asarray(num_messages, dtype=np.int64), kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        parts[r].append(int(i))
        load[r] += int(num_messages[i])
    return [sorted(p) for p in parts]


def datapoint_messages(datapoint) -> int:
    """Message count of a raw (un-tensorised) BugLab datapoint: the edges it lists, which is what a step's time follows."""
    try:
        edges = datapoint["graph"]["edges"]
        return int(sum(len(v) for v in edges.values())) if hasattr(edges, "values") else 1
    except Exception:
        return 1


def balanced_rank_share(data, rank: int, world: int, window_per_rank: int = 16, size_fn=datapoint_messages):
    """This rank's share of a stream that every rank iterates in the same order: the stream is cut in windows of
    `world * window_per_rank` datapoints and each window is split with `partition_by_messages` (greedy bin-packing by
    message count), so that ranks get about the same number of MESSAGES per step, not just the same number of graphs.
    Deterministic given the stream: every datapoint is taken by exactly one rank."""
    if world <= 1:
        yield from data
        return
    window: list = []

    def flush():
        parts = partition_by_messages([size_fn(d) for d in window], world)
        for i in parts[rank]:
            yield window[i]

    for d in data:
        if d is None:
            continue
        window.append(d)
        if len(window) == world * window_per_rank:
            yield from flush()
            window = []
    if window:
        yield from flush()


def global_batch_weight(local_graphs: int, device) -> float:
    """B_rank / B_total, the factor that makes the sum over ranks of per-rank losses
    (`loc.mean() + repair / B_rank`, reference gnn.py:251) equal the full-minibatch loss."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    t = torch.tensor([float(local_graphs)], device=device)
    dist.all_reduce(t)
    return float(local_graphs) / float(t.item())


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

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

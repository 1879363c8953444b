#!/usr/bin/env python
"""CPU oracle step time against the number of torch threads (which thread count the bench's cpu_baseline should use on a many-core host)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import torch

from buglab.data.collate import collate_samples
from buglab.data.synthetic import make_samples
from oracle import buglab_oracle as O

cfg = O.OracleConfig(hidden=128, num_layers=8, num_edge_types=16, dropout=0.2)
mb = collate_samples(make_samples(4, seed=123, num_nodes=2000, num_messages=10000, num_edge_types=16), 16)
params = O.init_params(cfg, seed=0)
print("cores", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
for nt in (8, 16, 32, 64, 128):
    if nt > (os.cpu_count() or 8):
        break
    torch.set_num_threads(nt)
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        O.forward_backward(params, mb, cfg, seed=i + 1)
        ts.append(time.perf_counter() - t0)
    print(f"threads {nt}: {min(ts[1:]):.2f} s per 4-graph forward+backward = {4 / min(ts[1:]):.2f} graphs/s", flush=True)

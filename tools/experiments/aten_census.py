#!/usr/bin/env python
"""Which torch-native (aten) kernels does one training step still launch, from where?  One profiled step of the bench's gnn-mlp workload
(--graphs 15: the reference's minibatch regime); prints every aten op that launched a device kernel with its input shapes and the
innermost Python frames of this repository."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import torch
from torch.profiler import ProfilerActivity, profile

from buglab.data.collate import collate_samples, to_device
from buglab.data.synthetic import make_samples
from buglab.models import hip_ops
from buglab.models.gnn import build_gnn_mlp_module
from buglab.runtime.optim import FlatAdam

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=15)
ap.add_argument("--hidden", type=int, default=128)
a = ap.parse_args()
hip_ops.load_library()
torch.manual_seed(0)
samples = make_samples(a.graphs, seed=1000, num_nodes=2000, num_messages=10000, num_edge_types=16)
mb = to_device(collate_samples(samples, 16), "cuda")
m = build_gnn_mlp_module(a.hidden, 8, 16, dropout_rate=0.2, embedder_dropout_rate=0.0).cuda().train()
opt = FlatAdam(m.parameters())
hip_ops.use_step_stream(torch.device("cuda"))


def step():
    opt.zero_grad()
    loss = m(**mb)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 and e.self_device_time_total <= 0:
        continue
    if e.self_device_time_total <= 0:
        continue
    frames = [f for f in (e.stack or []) if "buglab" in f or "bench" in f]
    key = (e.name, str(e.input_shapes)[:80], " <- ".join(f.split("/")[-1][:60] for f in frames[:3]))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += e.self_device_time_total
print(f"{'n':>3} {'us':>8}  op / shapes / where")
for (name, shapes, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:3d} {us:8.1f}  {name}  {shapes}\n{'':14s}{where}")
print("aten kernels per step:", sum(v[0] for v in rows.values()), " device us:", round(sum(v[1] for v in rows.values()), 1))

#!/usr/bin/env python
"""How fast are fp32 global atomics for row scatter-adds (E rows of D floats into N rows)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np, torch
from buglab.models import hip_ops as ops
N, E, D = 128000, 640000, 128
rng = np.random.default_rng(0)
tgt = np.sort(rng.integers(0, N, E)).astype(np.int32)
src = (tgt // 2000 * 2000 + rng.integers(0, 2000, E)).astype(np.int32)
x = torch.randn(E, 2 * D, device="cuda")
out = torch.zeros(N, D, device="cuda")
for name, idx, off in (("random-in-graph rows", src, 0), ("sorted rows", tgt, D)):
    i = torch.from_numpy(idx).cuda()
    for _ in range(2): ops.scatter_add_rows(x, off, D, i, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.scatter_add_rows(x, off, D, i, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms for {E*D/1e6:.0f}M fp32 atomics = {E*D/ms/1e9:.1f} G atomics/s, {E*D*4/ms/1e9:.2f} TB/s payload")
# plain copy of the same volume for reference
y = torch.empty(E, D, device="cuda")
for _ in range(2): y.copy_(x[:, :D])
torch.cuda.synchronize(); e0.record()
for _ in range(10): y.copy_(x[:, :D])
e1.record(); torch.cuda.synchronize()
print(f"strided copy [E,{D}]: {e0.elapsed_time(e1)/10:.3f} ms")

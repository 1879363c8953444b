#!/usr/bin/env python
"""How many host threads should the fp64 oracle of the parity tests use on the GPU box (256 hardware threads)?  The 8-graph
BASELINE-size case of tests/test_hip_parity.py, forward under no_grad and forward + backward, per torch thread count."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import torch

from oracle import buglab_oracle as O
from tests import helpers as Hh

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg, _, mb = Hh.make_case(B=B, n=2000, E=10000, T=16, H=128, layers=8, vocab=15000, C=40, degree="uniform", max_degree=512, seed=21,
                          msg_act_placement="aggregated")
p64 = {k: v.double() for k, v in O.init_params(cfg, seed=0).items()}
print("host threads", os.cpu_count(), "torch default", torch.get_num_threads(), flush=True)
for nt in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,16,32,64,0").split(",")]:
    torch.set_num_threads(nt if nt > 0 else (os.cpu_count() or 8))
    t = time.time()
    with torch.no_grad():
        O.forward_loss(p64, mb, cfg)
    t1 = time.time() - t
    t = time.time()
    O.forward_backward(p64, mb, cfg)
    print(f"B={B} threads {torch.get_num_threads():4d}: forward (no_grad) {t1:6.2f} s, forward + backward {time.time() - t:6.2f} s", flush=True)

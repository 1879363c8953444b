#!/usr/bin/env python
"""Golden trajectory of the optimiser row T1, produced by the REFERENCE's own pieces (build container only):
`optimizer()` and `LinearWarmupScheduler` from /root/reference/buglab/models/utils.py:51-66 driving `torch.optim.Adam`,
with the `clip_grad_norm_(…, 0.5)` the trainer is configured with (train.py:104), stepped the way ptgnn's trainer steps
them (optimiser step, then scheduler step, every minibatch).

    python tests/golden/make_golden_optim.py        # rewrites tests/golden/optim_trajectory.npz

Stand-ins: the third-party imports of utils.py (same list as make_golden.py); `AbstractScheduler` is ptgnn's empty base.
The fixture holds the initial parameters, the per-step gradients and the parameters after every step."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden as MG  # noqa: E402


def main():
    MG._install_stubs()
    import types

    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.path.insert(0, REF)
    from buglab.models.utils import LinearWarmupScheduler, optimizer  # noqa: reference code

    g = torch.Generator().manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(11, 5, generator=g)), torch.nn.Parameter(torch.randn(7, generator=g))]
    init = [p.detach().clone().numpy() for p in params]
    warmup, steps, lr, clip = 5, 12, 1e-2, 0.5
    opt = optimizer(params, lr=lr)
    sched = LinearWarmupScheduler(opt, num_warmup_steps=warmup)
    grads, after = [], []
    for k in range(steps):
        scale = 3.0 if k % 3 != 2 else 0.01  # clipped steps and un-clipped ones
        gk = [torch.randn(p.shape, generator=g) * scale for p in params]
        for p, gg in zip(params, gk):
            p.grad = gg.clone()
        torch.nn.utils.clip_grad_norm_(params, clip)
        opt.step()
        sched.step(epoch_idx=0, epoch_step=k)
        grads.append([x.numpy() for x in gk])
        after.append([p.detach().clone().numpy() for p in params])
    np.savez(os.path.join(OUT, "optim_trajectory.npz"), warmup=warmup, lr=lr, clip=clip, steps=steps,
             init0=init[0], init1=init[1],
             grads0=np.stack([x[0] for x in grads]), grads1=np.stack([x[1] for x in grads]),
             after0=np.stack([x[0] for x in after]), after1=np.stack([x[1] for x in after]))
    print("wrote optim_trajectory.npz:", steps, "steps; last lr factor", min(1.0, (steps - 1) / warmup))


if __name__ == "__main__":
    main()

"""Flat-buffer optimiser for the MI355X path.

The reference trains with `torch.optim.Adam(lr=1e-4)` (buglab/models/utils.py:51-52), gradient-norm
clipping at 0.5 and an 800-step linear warm-up (buglab/models/train.py:98-107, utils.py:55-66),
i.e. ~60 per-tensor kernels per step plus a host-synchronising `clip_grad_norm_`.  Here all
parameters live in ONE fp32 buffer and all gradients in another (the parameters' `.data`/`.grad`
are views into them), so a step is: [one RCCL all-reduce of the gradient buffer when data-parallel]
-> one squared-norm reduction -> one fused clip+Adam kernel.  Nothing returns to the host.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from buglab.models import hip_ops


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, clip_gradient_norm: float = 0.5,
                 num_warmup_steps: int = 800, betas=(0.9, 0.999), eps: float = 1e-8, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep every view 16-byte aligned
        self.numel = sum(sizes)
        self.flat_param = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.params, sizes):
            view = self.flat_param[off : off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_grad[off : off + p.numel()].view(p.shape)
            # opt in to the kernels' direct accumulation into .grad (hip_ops._direct_*): this optimiser joins the
            # side stream before it reads the gradients; parameters of any other optimiser keep plain autograd semantics
            p._bl_direct_grad = True
            off += n
        self.m = torch.zeros_like(self.flat_param)
        self.v = torch.zeros_like(self.flat_param)
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr, self.clip, self.warmup = lr, clip_gradient_norm, num_warmup_steps
        self.beta1, self.beta2 = betas
        self.eps = eps
        self.step_count = 0
        self.process_group = process_group

    def zero_grad(self):
        hip_ops.join_side_stream()
        self.flat_grad.zero_()

    def lr_at(self, step: int) -> float:
        """LambdaLR semantics of the reference's LinearWarmupScheduler (utils.py:55-66): the k-th
        optimiser step (k = 1, 2, ...) runs with factor min(1, (k - 1) / warmup)."""
        if self.warmup <= 0:
            return self.lr
        return self.lr * min(1.0, float(step - 1) / float(max(1, self.warmup)))

    def reduce_gradients(self, grad_weight: float = 1.0) -> float:
        """Data-parallel reduction: ONE all-reduce (sum) of the flat gradient buffer (RCCL over xGMI
        on GPUs, gloo in the CPU tests).  `grad_weight` = B_rank / B_total, this rank's share of the
        global minibatch, so that the sum equals the full-minibatch gradient.  Returns the factor
        still to be applied to the buffer (folded into the fused Adam kernel)."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            if grad_weight != 1.0:
                self.flat_grad.mul_(grad_weight)  # weights may differ per rank: scale locally, then sum
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.process_group)
            return 1.0
        return grad_weight

    def step(self, grad_weight: float = 1.0):
        hip_ops.join_side_stream()  # weight-gradient GEMMs accumulate into flat_grad on the side stream
        self.step_count += 1
        prescale = self.reduce_gradients(grad_weight)
        if self.flat_param.is_cuda:
            hip_ops.sqnorm(self.flat_grad, self.sqnorm)
            hip_ops.adam_clip_step(self.flat_param, self.flat_grad, self.m, self.v, self.sqnorm, prescale=prescale,
                                   clip=self.clip, lr=self.lr_at(self.step_count), beta1=self.beta1, beta2=self.beta2,
                                   eps=self.eps, step=self.step_count)
            hip_ops.invalidate_weight_packs()  # the kernel wrote the parameters behind autograd's version counters
        else:
            raise hip_ops.HipOpsUnavailable("FlatAdam.step: parameters are not on a ROCm device (no CPU fallback)")

    def grad_norm(self) -> float:
        """Global L2 norm of the last reduced gradient (host sync; diagnostics only)."""
        return float(self.sqnorm.sqrt())

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count}

    def load_state_dict(self, sd):
        hip_ops.invalidate_weight_packs()
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])

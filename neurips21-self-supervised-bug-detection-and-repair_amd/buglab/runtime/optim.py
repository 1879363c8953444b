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
    TAIL = 4  # floats appended to the gradient buffer and all-reduced with it: [graphs on this rank, rank had a minibatch, -, -]

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, clip_gradient_norm: float = 0.5,
                 num_warmup_steps: int = 800, betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, distributed: bool = True):
        self.distributed = distributed  # False: never all-reduce (a single-replica reference run inside a multi-rank job)
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep every view 16-byte aligned
        self.numel = sum(sizes)
        self.flat_param = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self._grad_and_tail = torch.zeros(self.numel + self.TAIL, dtype=torch.float32, device=dev)
        self.flat_grad = self._grad_and_tail[: self.numel]
        self.tail = self._grad_and_tail[self.numel :]
        self._tail_host = [torch.zeros(self.TAIL, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(self.TAIL)
                           for _ in range(2)]
        self._tail_event = [None, None]
        self._tail_slot = 0
        self._last_dp_step_counted = False
        off = 0
        self._span = {}  # id(param) -> (offset, padded size) in the flat buffers
        for p, n in zip(self.params, sizes):
            self._span[id(p)] = (off, n)
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
        # layer-wise buckets of the data-parallel gradient reduction (set_overlap_groups); None: one all-reduce per step
        self._buckets = None
        self._rest = None
        self._armed_B = None
        self._issued = 0
        self._bucket_left = None
        self._works = []
        self._comm_stream = None

    def zero_grad(self):
        hip_ops.join_side_stream()
        self._grad_and_tail.zero_()

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

        if self.distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
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

    # ---- data parallel: everything a step needs from the other ranks rides on the ONE gradient all-reduce ------------
    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica start from rank `src`'s parameters (and optimiser moments)."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            for t in (self.flat_param, self.m, self.v):
                dist.broadcast(t, src=src, group=self.process_group)
            hip_ops.invalidate_weight_packs()

    # ---- layer-wise buckets: the reduction of layer k's gradients runs while layers k-1 .. 1 are still in backward ------
    def set_overlap_groups(self, groups) -> bool:
        """`groups`: parameter lists in FORWARD order (one per message-passing layer), each contiguous in the flat buffer.
        With them, `begin_data_parallel_step` + the kernels' "backward of this layer has been launched" notifications
        (hip_ops.GRAD_READY_CALLBACK) start one all-reduce per group as soon as its gradients are complete -- in a FIXED
        order (last group first), on a communication stream of its own; `step_data_parallel` reduces what is left (the
        parameters outside the groups and the 4-float tail) and waits.  The plan depends on the model only, so every
        rank -- also one that has no minibatch and never runs backward -- issues the same collectives in the same order.
        Returns False (and keeps the single all-reduce) when a group is not contiguous."""
        buckets = []
        for g in groups:
            spans = sorted(self._span[id(p)] for p in g if id(p) in self._span)
            if not spans:
                continue
            for (o0, n0), (o1, _) in zip(spans, spans[1:]):
                if o0 + n0 != o1:
                    return False
            buckets.append((spans[0][0], spans[-1][0] + spans[-1][1], [id(p) for p in g if id(p) in self._span]))
        buckets.sort(key=lambda b: b[0])
        for (_, hi, _), (lo, _, _) in zip(buckets, buckets[1:]):
            if hi > lo:
                return False
        rest, pos = [], 0
        for lo, hi, _ in buckets:
            if lo > pos:
                rest.append((pos, lo))
            pos = hi
        rest.append((pos, self.numel + self.TAIL))  # never empty: the tail rides on the last range
        self._buckets, self._rest = buckets, rest
        self._bucket_of = {pid: i for i, (_, _, ids) in enumerate(buckets) for pid in ids}
        return True

    def _dp_active(self) -> bool:
        import torch.distributed as dist

        return bool(self.distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1)

    def begin_data_parallel_step(self, local_graphs: int) -> None:
        """Call after zero_grad() and BEFORE the forward pass of a data-parallel step (optional: without it, or without
        overlap groups, step_data_parallel does one all-reduce of the whole buffer)."""
        self._armed_B = None
        if self._buckets is None or not self._dp_active():
            return
        self._armed_B = int(local_graphs)
        self._issued = 0
        self._bucket_left = [len(ids) for _, _, ids in self._buckets]
        self._works = []
        if self._armed_B > 0:
            hip_ops.set_grad_ready_callback(self._on_layer_backward_launched)

    def abort_data_parallel_step(self) -> None:
        """Leave an armed step cleanly when forward / backward raised between `begin_data_parallel_step` and
        `step_data_parallel` (an out-of-memory minibatch, a caller that skips the step): the backward callback is
        disarmed -- a later backward must not issue collectives nobody expects -- and the collectives of THIS step's plan
        that were not issued yet are issued and waited for, so that the peers, which issue the same plan, are not left
        hanging in it while the exception travels up on this rank.  The reduced values are meaningless; no update follows."""
        hip_ops.set_grad_ready_callback(None)
        if self._armed_B is None:
            return
        try:
            self._issue_ready(everything=True)
            for lo, hi in self._rest:
                self._all_reduce_range(lo, hi)
            for w in self._works:
                w.wait()
        finally:
            self._works = []
            self._armed_B = None

    def _on_layer_backward_launched(self, params) -> None:
        for p in params:
            b = self._bucket_of.get(id(p))
            if b is not None:
                self._bucket_left[b] -= 1
        self._issue_ready()

    def _issue_ready(self, everything: bool = False) -> None:
        # canonical order: last group first; a group is issued only after every later group has been
        n = len(self._buckets)
        while self._issued < n:
            b = n - 1 - self._issued
            if not everything and self._bucket_left[b] > 0:
                return
            lo, hi, _ = self._buckets[b]
            self._all_reduce_range(lo, hi)
            self._issued += 1

    def _all_reduce_range(self, lo: int, hi: int) -> None:
        import torch.distributed as dist

        seg = self._grad_and_tail[lo:hi]
        B = self._armed_B
        if seg.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            # the gradients of this range are complete once everything launched so far on the training stream AND on the
            # free-running side stream (weight-gradient GEMMs) has finished
            comm = self._comm_stream
            ev = torch.cuda.Event()
            ev.record()
            comm.wait_event(ev)
            side = hip_ops.side_stream_if_any()
            if side is not None:
                ev2 = torch.cuda.Event()
                ev2.record(side)
                comm.wait_event(ev2)
            with torch.cuda.stream(comm):
                if B > 0 and lo < self.numel:
                    self._grad_and_tail[lo:min(hi, self.numel)].mul_(float(B))
                self._works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True))
        else:
            if B > 0 and lo < self.numel:
                self._grad_and_tail[lo:min(hi, self.numel)].mul_(float(B))
            self._works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True))

    def step_data_parallel(self, local_graphs: int) -> None:
        """One optimiser step of a data-parallel run.  Each rank calls it EVERY step, with the number of graphs of the
        minibatch it just back-propagated (0, with an untouched zero gradient buffer, when its loader is exhausted).
        The gradient buffer is scaled by that count and all-reduced together with a 4-float tail carrying the count
        and a "had a minibatch" flag: one collective per step, no host synchronisation.  The fused clip+Adam kernel
        divides by the global count it finds in the tail (on the device) and does nothing when that count is zero."""
        import torch.distributed as dist

        hip_ops.join_side_stream()
        B = int(local_graphs)
        self.tail.zero_()
        if B > 0:
            self.tail[0].fill_(float(B))  # fill kernels: a host tensor would have to be copied from pageable memory (blocking)
            self.tail[1].fill_(1.0)
        if self._armed_B is not None:
            # bucketed form: the layer groups not reduced during backward (none, on a rank without a minibatch) in the same
            # fixed order, then the ranges outside the groups with the tail; then wait for all of them
            assert self._armed_B == B, "begin_data_parallel_step / step_data_parallel disagree on the number of graphs"
            hip_ops.set_grad_ready_callback(None)
            self._issue_ready(everything=True)
            for lo, hi in self._rest:
                self._all_reduce_range(lo, hi)
            for w in self._works:
                w.wait()
            self._works = []
            self._armed_B = None
        else:
            if B > 0:
                self.flat_grad.mul_(float(B))
            if self._dp_active():
                dist.all_reduce(self._grad_and_tail, op=dist.ReduceOp.SUM, group=self.process_group)
        self.step_count += 1
        self._last_dp_step_counted = True
        self._apply_update_data_parallel()
        hip_ops.invalidate_weight_packs()
        # the tail of THIS step goes to pinned memory asynchronously; `previous_step_was_idle` reads it one step later
        slot = self._tail_slot
        self._tail_host[slot].copy_(self.tail, non_blocking=True)
        ev = None
        if self.tail.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._tail_event[slot] = ev
        self._tail_slot = slot ^ 1
        self._dp_steps = getattr(self, "_dp_steps", 0) + 1

    def _apply_update_data_parallel(self) -> None:
        """Squared norm + fused clip / Adam on the device (the CPU tests of the protocol override this)."""
        if not self.flat_param.is_cuda:
            raise hip_ops.HipOpsUnavailable("FlatAdam.step_data_parallel: parameters are not on a ROCm device (no CPU fallback)")
        hip_ops.sqnorm(self.flat_grad, self.sqnorm)
        hip_ops.adam_clip_step_dp(self.flat_param, self.flat_grad, self.m, self.v, self.sqnorm, self.tail, clip=self.clip,
                                  lr=self.lr_at(self.step_count), beta1=self.beta1, beta2=self.beta2, eps=self.eps, step=self.step_count)

    def previous_step_was_idle(self) -> bool:
        """True when NO rank had a minibatch in the most recent `step_data_parallel` (the epoch is over for everyone).
        Reads the pinned copy made by that step: by the time the next minibatch has been fetched the copy is long
        complete, so this does not stall the device queue.  The idle step is taken back from the step counter."""
        if getattr(self, "_dp_steps", 0) == 0:
            return False
        slot = self._tail_slot ^ 1
        ev = self._tail_event[slot]
        if ev is not None:
            ev.synchronize()
        idle = float(self._tail_host[slot][1]) == 0.0
        if idle and self._last_dp_step_counted:
            self.step_count -= 1  # nothing was updated (the kernel saw a zero global count)
            self._last_dp_step_counted = False
        return idle

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

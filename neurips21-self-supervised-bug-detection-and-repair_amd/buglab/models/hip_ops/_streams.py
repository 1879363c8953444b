"""Stream policy and live kernel timing of buglab.models.hip_ops.

* side stream: the weight-gradient GEMMs run next to the input-gradient chain of the same layer and, for parameters that
  opted in to direct gradient accumulation, keep running behind the main chain until `join_side_stream()`;
* step stream: the training step's dependent chain on a high-priority stream (`use_step_stream`);
* `KernelTimer` / `_timed`: HIP events around every launch for bench.py's roofline tables.
All mutable state of the three lives HERE (one module, one copy); `hip_ops.USE_SIDE_STREAM = ...` on the package is forwarded
to this module (hip_ops/__init__.py::_HipOpsModule)."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int32, c_int64
from typing import Optional

import torch

from ._lib import _check, load_library

# ------------------------------------------------------------------------------------------------
# optional live kernel timing (bench.py): HIP events recorded on the launch stream around each GEMM
class KernelTimer:
    """`with KernelTimer() as t:` brackets every bl_gemm_* launch with a pair of HIP events on the
    stream the kernel is launched on (torch's current stream is the one handed to the C ABI).
    `t.summary()` (after a device sync) -> {kind: {"launches", "ms", "flop"}}."""

    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        KernelTimer.active = self
        lib = load_library()
        lib.bl_prof_reset()
        lib.bl_prof_enable(1)  # kernels launched inside the fused per-layer calls are timed on the C side
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        load_library().bl_prof_enable(0)

    def summary(self):
        """{kind: {launches, ms, flop, overlapped}}.  `overlapped` kinds were launched while a kernel
        of the same layer ran on the side stream: their event spans share the GPU and must not be
        read as exclusive kernel time (the enclosing "*_pair" span is the exclusive one)."""
        out = {}
        for kind, flop, e0, e1, overlapped, nbytes in self.records:
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "flop": 0.0, "overlapped": False})
            d["overlapped"] = d["overlapped"] or overlapped
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flop"] += flop
            if nbytes:
                d["bytes"] = d.get("bytes", 0.0) + nbytes  # algorithmic bytes of a memory-bound kind
        lib = load_library()
        for k in range(lib.bl_prof_num_kinds()):
            ms, flop, n, ov = ctypes.c_double(), ctypes.c_double(), c_int64(), c_int32()
            _check(lib.bl_prof_read(k, ctypes.byref(ms), ctypes.byref(flop), ctypes.byref(n), ctypes.byref(ov)), "bl_prof_read")
            if n.value:
                name = lib.bl_prof_kind_name(k).decode()
                d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flop": 0.0, "overlapped": False})  # (a kind may also be timed from Python)
                d["launches"] += int(n.value)
                d["ms"] += ms.value
                d["flop"] += flop.value
                d["overlapped"] = d["overlapped"] or bool(ov.value)
                nbytes = float(lib.bl_prof_read_bytes(k))
                if nbytes:
                    d["bytes"] = d.get("bytes", 0.0) + nbytes
        return out


_overlap_depth = 0
_free_running = False  # weight-gradient GEMMs of earlier layers may still be running on the side stream


class _timed:
    def __init__(self, kind, flop, span=False, nbytes=0.0):
        self.t = KernelTimer.active
        self.kind, self.flop, self.span, self.nbytes = kind, flop, span, nbytes

    def __enter__(self):
        if self.t is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.t is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.t.records.append((self.kind, self.flop, self.e0, e1, (not self.span) and (_overlap_depth > 0 or _free_running), self.nbytes))


# ------------------------------------------------------------------------------------------------
# side stream: weight-gradient GEMMs run next to the input-gradient chain of the same layer (both
# only read the node gradient), so one kernel's prologue / epilogue / last-round tail is filled by
# the other kernel's workgroups.  BL_SIDE_STREAM=0 disables.
_side_streams = {}
USE_SIDE_STREAM = os.environ.get("BL_SIDE_STREAM", "1") != "0"
# Weight gradients of the message-passing layers are accumulated (fp32 atomics in the kernel)
# straight into `param.grad` for parameters whose owner OPTED IN (`param._bl_direct_grad = True`, set by
# FlatAdam, which pre-binds every .grad to a view of its flat gradient buffer), on the side stream, WITHOUT
# joining at the end of the layer's backward: the side stream runs one weight-gradient GEMM after the other
# behind the main chain and is joined once, by `join_side_stream()`, before the gradients are consumed
# (FlatAdam.zero_grad / .step).  Parameters of any other optimiser get ordinary autograd gradients, complete
# when backward() returns (the side stream is joined inside the layer's backward).
DIRECT_PARAM_GRAD = os.environ.get("BL_DIRECT_GRAD", "1") != "0"


_held_for_side_stream: list = []  # tensors the free-running side-stream GEMMs read: kept alive until the join


# Priority of the side stream that carries the weight-gradient GEMMs (lower number = higher priority; out-of-range values are
# mapped to the nearest valid one).  BL_SIDE_STREAM_PRIORITY: A/B knob.
SIDE_STREAM_PRIORITY = int(os.environ.get("BL_SIDE_STREAM_PRIORITY", "0"))


def _new_side_stream():
    return torch.cuda.Stream(priority=SIDE_STREAM_PRIORITY) if SIDE_STREAM_PRIORITY != 0 else torch.cuda.Stream()


# The training step's dependent chain (forward, the backward's input-gradient chain, clip + Adam) runs on a HIGH-priority stream,
# the weight-gradient GEMMs that run beside it on a normal-priority one (the chip has two levels: 0 and -1): when both have
# workgroups to place, the chain's kernels get the CUs first and the weight gradients fill what they leave -- 17.50 -> 17.33 ms
# per step on one box, two A/B pairs (profiles/r04o_*).  BUGLAB_STEP_STREAM_PRIORITY=0 keeps the caller's stream.
_step_streams = {}


def use_step_stream(device=None):
    """Make a high-priority stream the current stream of this thread (once per device; later calls re-select it).  Work queued on
    the previous current stream is waited for.  -> the stream, or None when switched off / no GPU."""
    if not torch.cuda.is_available() or os.environ.get("BUGLAB_STEP_STREAM_PRIORITY", "-1") == "0":
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _step_streams.get(key)
    if st is None:
        st = _step_streams[key] = torch.cuda.Stream(dev, priority=int(os.environ.get("BUGLAB_STEP_STREAM_PRIORITY", "-1")))
    cur = torch.cuda.current_stream(dev)
    if cur != st:
        st.wait_stream(cur)
        torch.cuda.set_stream(st)
    return st


def join_side_stream():
    """Make the current stream wait for every weight-gradient GEMM still running on the side stream.  What those GEMMs
    read is released only now, i.e. behind the wait in this stream's order: the allocator may hand the blocks to the next
    step at once (with `record_stream` instead they stayed unusable until their events completed, and a run settled
    at ~100 GiB reserved for a 6 GiB peak)."""
    global _free_running
    if torch.cuda.is_available():
        key = torch.cuda.current_device()
        if key in _side_streams:
            torch.cuda.current_stream().wait_stream(_side_streams[key])
    _held_for_side_stream.clear()
    _free_running = False


def _opted_in_for_direct_grad(param) -> bool:
    """THE CONTRACT of direct gradient accumulation (every Function of this module that owns parameters: the message-passing
    layers, gather_linear, mlp_score, localization_scores, rowdot, the relational attention's bias tables): a parameter
    whose owner set `param._bl_direct_grad = True` and bound `param.grad` to a preallocated fp32 buffer (FlatAdam does both for
    the parameters it owns, zeroing the flat buffer in zero_grad()) gets its gradient ADDED INTO `param.grad` by the kernels,
    and backward returns None for it.  Consequences: `torch.autograd.grad(...)` sees no gradient for such a parameter and
    tensor hooks registered on it would never fire -- so a parameter that has hooks (or post-accumulate-grad hooks) is treated
    as not opted in and receives its gradient through autograd as usual.  Parameters without the flag always take that
    path."""
    if not (DIRECT_PARAM_GRAD and getattr(param, "_bl_direct_grad", False)):
        return False
    if getattr(param, "_backward_hooks", None) or getattr(param, "_post_accumulate_grad_hooks", None):
        return False
    return True


def _direct_small(param):
    """.grad of a small (bias / LayerNorm) parameter when the kernels may accumulate into it directly
    (FlatAdam's flat gradient buffer): no zero-fill, no autograd accumulation kernel.  Contract: _opted_in_for_direct_grad."""
    g = getattr(param, "grad", None)
    if _opted_in_for_direct_grad(param) and g is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous():
        return g
    return None


def _direct_grad_target(param):
    g = getattr(param, "grad", None)
    if (USE_SIDE_STREAM and _opted_in_for_direct_grad(param) and g is not None and g.is_cuda
            and g.dtype == torch.float32 and g.is_contiguous()):
        return g
    return None


class _on_side_stream:
    def __init__(self, device):
        self.enabled = USE_SIDE_STREAM
        if self.enabled:
            key = torch.cuda.current_device()
            if key not in _side_streams:
                _side_streams[key] = _new_side_stream()
            self.side = _side_streams[key]
            self.main = torch.cuda.current_stream()

    def __enter__(self):
        global _overlap_depth
        if self.enabled:
            _overlap_depth += 1
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)

    def join(self):
        global _overlap_depth
        if self.enabled:
            self.main.wait_stream(self.side)
            _overlap_depth -= 1

    def detach(self, *tensors):
        """Leave the side-stream work running: the tensors it reads stay referenced until `join_side_stream()`."""
        global _overlap_depth
        if self.enabled:
            _held_for_side_stream.extend(t for t in tensors if t is not None)
            _overlap_depth -= 1



def mark_free_running(*held) -> None:
    """Weight-gradient GEMMs were left running on the side stream: what they read (`held`) stays referenced until the join."""
    global _free_running
    _free_running = True
    _held_for_side_stream.extend(t for t in held if t is not None)


def side_stream_for_current_device():
    """The side stream of the current device (created on first use), or None when the side stream is switched off."""
    if not USE_SIDE_STREAM:
        return None
    key = torch.cuda.current_device()
    if key not in _side_streams:
        _side_streams[key] = _new_side_stream()
    return _side_streams[key]


def side_stream_if_any():
    return _side_streams.get(torch.cuda.current_device()) if torch.cuda.is_available() else None


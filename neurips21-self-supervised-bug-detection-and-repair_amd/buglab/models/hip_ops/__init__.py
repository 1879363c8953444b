"""ctypes binding of libbuglab_hip.so (C ABI in include/buglab_hip.h) + torch.autograd wrappers.

This is the ONLY place the HIP kernels are called from.  There is no CPU fallback: importing works
anywhere (so host-side code can be tested without a GPU), but the first call that needs a kernel
raises `HipOpsUnavailable` if the shared library is missing or the tensors are not on a ROCm device.

PyTorch is used for device memory, streams and autograd bookkeeping only; every FLOP of the
message-passing layers and of the scoring heads runs in the kernels under csrc/.
"""
from __future__ import annotations

import ctypes
import os
import sys
import types
from ctypes import POINTER, Structure, c_float, c_int8, c_int32, c_int64, c_uint32, c_void_p
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch

from . import _lib as _lib_module
from . import _streams
from ._lib import *  # noqa: F401,F403  (ACT_* codes, structures, Dropout, load_library, HipOpsUnavailable, ...)
from ._lib import _ACTS, _SIGNATURES, _check, _f32, _i32, _p, _req, _rows, _stream  # noqa: F401
from ._streams import (KernelTimer, _direct_grad_target, _direct_small, _on_side_stream, _opted_in_for_direct_grad, _timed,  # noqa: F401
                       join_side_stream, side_stream_if_any, use_step_stream)

# Debug tap for the parity tests: when set to a list, every message-passing layer's forward appends its
# winner table (int32 [N, Dm]: id of the message that won each channel's max at each node, -1 = none).
WINNER_SINK: Optional[list] = None

# ------------------------------------------------------------------------------------------------
# raw (non-autograd) entry points
def gemm_rows(sources, b, M, N, *, b_is_nk=False, b_group_stride=0, ldb=None, bias=None, group_ptr=None, group_w=None,
              G=1, act=ACT_NONE, drop: Dropout = NO_DROPOUT, out=None):
    rows, K = _rows(sources)
    _f32(b, "b")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=b.device)
    if M == 0:
        return out
    kind = ("gemm_rows_nk" if b_is_nk else "gemm_rows") + ("_grouped" if group_ptr is not None else "")
    with _timed(kind, 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_rows(
                ctypes.byref(rows), b.data_ptr(), int(b_group_stride), int(ldb if ldb is not None else b.shape[-1]), int(b_is_nk),
                _p(bias), _p(group_ptr), _p(group_w), int(G), int(M), int(N), int(K), int(act), drop.c(), out.data_ptr(),
                out.stride(0), _stream()),
            "bl_gemm_rows")
    return out


# ------------------------------------------------------------------------------------------------
# fp32-accurate GEMM on the bf16 matrix cores (csrc/bl_gemm_x6.hip)
GEMM_MODE = os.environ.get("BL_GEMM_MODE", "bf16x6")  # "bf16x6" | "fp32"
WGRAD_X6 = os.environ.get("BL_WGRAD_X6", "1") != "0"   # bf16x6 weight gradient of the message layers


def pack_bf16x3(x: torch.Tensor) -> torch.Tensor:
    """fp32 [R, D] -> packed int16 [R, 3 * D]: per row the three bf16 planes [hi x D | mid x D | lo x D]."""
    _f32(x, "x")
    R, D = x.shape
    out = torch.empty((R, 3 * D), dtype=torch.int16, device=x.device)
    _check(load_library().bl_pack_bf16x3(x.data_ptr(), x.stride(0), R, D, out.data_ptr(), _stream()), "bl_pack_bf16x3")
    return out


def pack_weights_x6(w: torch.Tensor, w_is_kn: bool) -> torch.Tensor:
    """fp32 weights -> the tiled packed B operand of gemm_rows_x6 (int16 [G, tiles * stages * 12288]).
    w is [G, K, N] when w_is_kn (C = A @ w[g]) or [G, N, K] (C = A @ w[g]^T)."""
    _f32(w, "w")
    G, K, N = (w.shape[0], w.shape[1], w.shape[2]) if w_is_kn else (w.shape[0], w.shape[2], w.shape[1])
    out = torch.empty((G, ((N + 127) // 128) * (K // 32) * 12288), dtype=torch.int16, device=w.device)
    _check(load_library().bl_pack_weights_x6(w.data_ptr(), G, K, N, 1 if w_is_kn else 0, out.data_ptr(), _stream()), "bl_pack_weights_x6")
    return out


def rows_x6w_ok(N: int, K: int) -> bool:
    """Shapes the wide row GEMM takes (bl_gemm_rows_x6w_ok: N a multiple of 256, K of 64)."""
    return bool(load_library().bl_gemm_rows_x6w_ok(int(N), int(K)))


def pack_weights_x6w(w: torch.Tensor, w_is_kn: bool) -> torch.Tensor:
    """fp32 weights -> the weight image of gemm_rows_x6(..., wide=True) (bl_pack_weights_x6w; same shapes as pack_weights_x6)."""
    _f32(w, "w")
    G, K, N = (w.shape[0], w.shape[1], w.shape[2]) if w_is_kn else (w.shape[0], w.shape[2], w.shape[1])
    lib = load_library()
    out = torch.empty((G, int(lib.bl_packed_weight_elems_x6w(1, K, N))), dtype=torch.int16, device=w.device)
    _check(lib.bl_pack_weights_x6w(w.data_ptr(), G, K, N, 1 if w_is_kn else 0, out.data_ptr(), _stream()), "bl_pack_weights_x6w")
    return out


def gemm_rows_x6(sources, bp, M, N, *, group_ptr=None, group_w=None, G=1, win_bits=None, kind="gemm_rows_x6", bias=None, act=None,
                 drop: "Dropout" = None, wide: bool = False):
    """sources: [(packed int16 [*, 3*width], row index or None, width)]; bp: pack_weights_x6 output [G, *];
    win_bits: segment_max's per-row routing bitmask -> the routed (winner-masked) left operand;
    bias / act / drop: the epilogue drop(act(. + bias)) of bl_gemm_rows_x6_epi."""
    r = bl_rows_packed_t()
    K = 0
    for j, (xp, idx, width) in enumerate(sources):
        _req(xp, torch.int16, f"packed source {j}")
        r.xp[j] = xp.data_ptr()
        r.idx[j] = _i32(idx).data_ptr() if idx is not None else None
        r.width[j] = width
        K += width
    r.nsrc = len(sources)
    out = torch.empty((M, N), dtype=torch.float32, device=bp.device)
    if M == 0:
        return out
    if bias is not None or act is not None or drop is not None:
        with _timed(kind + "_epi", 2.0 * M * N * K):
            _check(
                load_library().bl_gemm_rows_x6_epi(ctypes.byref(r), _req(bp, torch.int16, "bp").data_ptr(), int(bp.stride(0)), _p(group_ptr),
                                                   _p(group_w), int(G), int(M), int(N), int(K), _p(bias), int(act or ACT_NONE),
                                                   (drop or NO_DROPOUT).c(), out.data_ptr(), out.stride(0), _stream()),
                "bl_gemm_rows_x6_epi")
        return out
    if wide:  # bp = pack_weights_x6w(...): the 128 x 256-tile kernel (bit-identical results)
        with _timed(kind + ("_grouped" if group_ptr is not None else ""), 2.0 * M * N * K):
            _check(
                load_library().bl_gemm_rows_x6w(ctypes.byref(r), _p(win_bits), win_bits.stride(0) if win_bits is not None else 0,
                                                _req(bp, torch.int16, "bp").data_ptr(), int(bp.stride(0)), _p(group_ptr), _p(group_w),
                                                int(G), int(M), int(N), int(K), out.data_ptr(), out.stride(0), _stream()),
                "bl_gemm_rows_x6w")
        return out
    with _timed(kind + ("_grouped" if group_ptr is not None else ""), 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_rows_x6(ctypes.byref(r), _p(win_bits), win_bits.stride(0) if win_bits is not None else 0,
                                           _req(bp, torch.int16, "bp").data_ptr(), int(bp.stride(0)), _p(group_ptr), _p(group_w),
                                           int(G), int(M), int(N), int(K), out.data_ptr(), out.stride(0), _stream()),
            "bl_gemm_rows_x6")
    return out


# ------------------------------------------------------------------------------------------------
# f16x3: fp32-accurate GEMMs on the fp16 matrix cores (csrc/bl_gemm_h3.hip) -- two fp16 planes per operand, three MFMA terms,
# power-of-two tensor scales.  H3_ROW_SCALE: layer inputs (|h| <= 1.25 after tanh x dropout; embedding rows), H3_W_SCALE: weights.
H3_ROW_SCALE = 256.0
H3_W_SCALE = 64.0


def pack_f16x2(x: torch.Tensor, scale: float = H3_ROW_SCALE, amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R, D] fp32 -> [R, 2 D] int16 (hi plane | lo plane of x * scale; amax: device float with max |x| -> the scale is derived on
    the device, the consumer GEMM takes the same tensor as `a_amax` / `g_amax`)."""
    _f32(x, "x")
    R, D = x.shape
    out = torch.empty((R, 2 * D), dtype=torch.int16, device=x.device)
    _check(load_library().bl_pack_f16x2(x.data_ptr(), x.stride(0), int(R), int(D), int(D), 0, float(scale), _p(amax), out.data_ptr(), _stream()),
           "bl_pack_f16x2")
    return out


def h3_saturation_events(reset: bool = False) -> int:
    """f16x2 packing threads that had to saturate a finite value since the last reset on the current device (synchronises): 0 in a
    healthy run -- a layer input beyond +-255.9 or a weight beyond +-1023 would count (bl_h3_saturation_events)."""
    return int(load_library().bl_h3_saturation_events(1 if reset else 0))


def amax(x: torch.Tensor) -> torch.Tensor:
    """device float [1] = max |x| (bl_amax)"""
    _f32(x, "x")
    out = torch.zeros((1,), dtype=torch.float32, device=x.device)
    _check(load_library().bl_amax(x.data_ptr(), int(x.numel()), out.data_ptr(), _stream()), "bl_amax")
    return out


def pack_weights_h3(w: torch.Tensor, w_is_kn: bool, scale: float = H3_W_SCALE) -> torch.Tensor:
    """w [G, K, N] (w_is_kn) or [G, N, K] -> tiled f16x2 image [G, *] int16 (bl_pack_weights_h3)"""
    _f32(w, "w")
    G, K, N = (w.shape[0], w.shape[1], w.shape[2]) if w_is_kn else (w.shape[0], w.shape[2], w.shape[1])
    lib = load_library()
    out = torch.empty((G, int(lib.bl_packed_weight_elems_h3(1, K, N))), dtype=torch.int16, device=w.device)
    _check(lib.bl_pack_weights_h3(w.data_ptr(), G, K, N, 1 if w_is_kn else 0, float(scale), out.data_ptr(), _stream()), "bl_pack_weights_h3")
    return out


def _rows_packed_h(sources):
    r = bl_rows_packed_t()
    K = 0
    for j, (xp, idx, width) in enumerate(sources):
        _req(xp, torch.int16, f"packed source {j}")
        r.xp[j] = xp.data_ptr()
        r.idx[j] = _i32(idx).data_ptr() if idx is not None else None
        r.width[j] = width
        K += width
    r.nsrc = len(sources)
    return r, K


def gemm_rows_h3(sources, bp, M, N, *, out_scale, group_ptr=None, group_w=None, G=1, win_bits=None, a_amax=None, kind="gemm_rows_h3"):
    """sources: [(pack_f16x2 rows, row index or None, width)]; bp: pack_weights_h3 image; out_scale = 1 / (row scale x weight scale)"""
    r, K = _rows_packed_h(sources)
    out = torch.empty((M, N), dtype=torch.float32, device=bp.device)
    if M == 0:
        return out
    with _timed(kind + ("_grouped" if group_ptr is not None else ""), 2.0 * M * N * K):
        _check(load_library().bl_gemm_rows_h3(ctypes.byref(r), _p(win_bits), int(win_bits.stride(0)) if win_bits is not None else 0,
                                              _req(bp, torch.int16, "bp").data_ptr(), int(bp.stride(0)), _p(group_ptr), _p(group_w), int(G),
                                              int(M), int(N), int(K), float(out_scale), _p(a_amax), out.data_ptr(), out.stride(0), _stream()),
               "bl_gemm_rows_h3")
    return out


def gemm_wgrad_h3(sources, g_packed, M, N, gw, *, out_scale, g_idx=None, win_bits=None, g_amax=None, gw_group_stride=0, group_ptr=None,
                  group_w=None, G=1):
    """gw[g] += out_scale * rows(sources)^T . G rows (g_idx gather, win_bits routing): bl_gemm_wgrad_h3"""
    r, K = _rows_packed_h(sources)
    if M == 0:
        return gw
    with _timed("gemm_wgrad_h3", 2.0 * M * N * K):
        _check(load_library().bl_gemm_wgrad_h3(ctypes.byref(r), _req(g_packed, torch.int16, "g_packed").data_ptr(), _p(g_idx), _p(win_bits),
                                               int(win_bits.stride(0)) if win_bits is not None else 0, _p(group_ptr), _p(group_w), int(G), int(M),
                                               int(N), int(K), float(out_scale), _p(g_amax), gw.data_ptr(), int(gw_group_stride), int(gw.stride(-2)),
                                               _stream()), "bl_gemm_wgrad_h3")
    return gw


def _rows_packed(sources):
    r = bl_rows_packed_t()
    K = 0
    for j, (xp, idx, width) in enumerate(sources):
        _req(xp, torch.int16, f"packed source {j}")
        r.xp[j] = xp.data_ptr()
        r.idx[j] = _i32(idx).data_ptr() if idx is not None else None
        r.width[j] = width
        K += width
    r.nsrc = len(sources)
    return r, K


def gemm_wgrad_routed_x6(sources, g_node_packed, node_of_row, win_bits, M, N, gw, *, gw_group_stride=0, group_ptr=None, group_w=None, G=1):
    """bf16x6 weight gradient of the routed (max-aggregated) messages; accumulates into gw."""
    rows, K = _rows_packed(sources)
    if M == 0:
        return gw
    with _timed("gemm_wgrad_routed_x6", 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_wgrad_routed_x6(ctypes.byref(rows), _req(g_node_packed, torch.int16, "g_node_packed").data_ptr(),
                                                   _i32(node_of_row).data_ptr(), _i32(win_bits).data_ptr(), win_bits.stride(0),
                                                   _p(group_ptr), _p(group_w), int(G), int(M), int(N), int(K),
                                                   _f32(gw).data_ptr(), int(gw_group_stride), int(gw.shape[-1]), _stream()),
            "bl_gemm_wgrad_routed_x6")
    return gw


def gemm_wgrad_x6(sources, g_packed, M, N, gw, *, g_idx=None, gw_group_stride=0, group_ptr=None, group_w=None, G=1):
    """gw[g] += rows(sources)^T . g_packed[(g_idx[r] or r)] from bf16x3-packed operands (no routing): the weight gradient of a
    plain Linear.  sources as in gemm_rows_x6; g_packed int16 [*, 3 N]."""
    r, K = _rows_packed(sources)
    if M == 0:
        return gw
    with _timed("gemm_wgrad_x6", 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_wgrad_x6(ctypes.byref(r), _req(g_packed, torch.int16, "g_packed").data_ptr(), _p(g_idx), _p(group_ptr),
                                            _p(group_w), int(G), int(M), int(N), int(K), _f32(gw, "gw").data_ptr(), int(gw_group_stride),
                                            int(gw.shape[-1]), _stream()),
            "bl_gemm_wgrad_x6")
    return gw


def x6_ok(*dims) -> bool:
    return GEMM_MODE == "bf16x6" and all(d % 32 == 0 for d in dims)


def gemm_rows_routed(g_node, node_of_row, winner, b, M, N, *, b_group_stride=0, ldb=None, group_ptr=None, group_w=None, G=1):
    """C[r, :] = (g_node[node_of_row[r]] masked to the entries row r won) . B_g^T  (include/buglab_hip.h)."""
    rows, K = _rows([(g_node, node_of_row)])
    out = torch.empty((M, N), dtype=torch.float32, device=b.device)
    if M == 0:
        return out
    with _timed("gemm_rows_nk_routed", 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_rows_routed(ctypes.byref(rows), _i32(winner).data_ptr(), winner.stride(0), _f32(b).data_ptr(),
                                               int(b_group_stride), int(ldb if ldb is not None else b.shape[-1]), _p(group_ptr),
                                               _p(group_w), int(G), int(M), int(N), int(K), out.data_ptr(), out.stride(0), _stream()),
            "bl_gemm_rows_routed")
    return out


def gemm_wgrad_routed(sources, g_node, node_of_row, winner, M, N, gw, *, gw_group_stride=0, group_ptr=None, group_w=None, G=1):
    rows, K = _rows(sources)
    if M == 0:
        return gw
    with _timed("gemm_wgrad_routed", 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_wgrad_routed(ctypes.byref(rows), _f32(g_node).data_ptr(), g_node.stride(0),
                                                _i32(node_of_row).data_ptr(), _i32(winner).data_ptr(), winner.stride(0),
                                                _p(group_ptr), _p(group_w), int(G), int(M), int(N), int(K), _f32(gw).data_ptr(),
                                                int(gw_group_stride), int(gw.shape[-1]), _stream()),
            "bl_gemm_wgrad_routed")
    return gw


def gemm_wgrad(sources, g_c, M, N, gw, *, gw_group_stride=0, group_ptr=None, group_w=None, G=1):
    rows, K = _rows(sources)
    _f32(g_c, "g_c")
    _f32(gw, "gw")
    if M == 0:
        return gw
    with _timed("gemm_wgrad" + ("_grouped" if group_ptr is not None else ""), 2.0 * M * N * K):
        _check(
            load_library().bl_gemm_wgrad(ctypes.byref(rows), g_c.data_ptr(), g_c.stride(0), _p(group_ptr), _p(group_w), int(G),
                                         int(M), int(N), int(K), gw.data_ptr(), int(gw_group_stride), int(gw.shape[-1]), _stream()),
            "bl_gemm_wgrad")
    return gw


def segment_max(x, seg_ptr, seg_items, nseg, act=ACT_NONE, ln=None, eps=1e-5, want_dact=False, want_bits=False, seg_order=None):
    """-> (out [nseg, D], arg int32 [nseg, D], ln_out | None, mean | None, rstd | None[, dact][, winbits])

    winbits: int32 [items, ceil(D/32)], bit d of row i set iff item i won channel d of its segment
    (every item must belong to exactly one segment)."""
    _f32(x, "x")
    D = x.shape[1]
    dev = x.device
    out = torch.empty((nseg, D), dtype=torch.float32, device=dev)
    arg = torch.empty((nseg, D), dtype=torch.int32, device=dev)
    ln_out = mean = rstd = None
    if ln is not None:
        ln_out = torch.empty((nseg, D), dtype=torch.float32, device=dev)
        mean = torch.empty((nseg,), dtype=torch.float32, device=dev)
        rstd = torch.empty((nseg,), dtype=torch.float32, device=dev)
    dact = torch.empty((nseg, D), dtype=torch.float32, device=dev) if want_dact else None
    bits = torch.empty((x.shape[0], (D + 31) // 32), dtype=torch.int32, device=dev) if want_bits else None
    _check(
        load_library().bl_segment_max_fwd(x.data_ptr(), x.stride(0), _i32(seg_ptr).data_ptr(), _p(seg_items), int(nseg), int(D),
                                          int(act), out.data_ptr(), arg.data_ptr(), _p(ln[0]) if ln else None,
                                          _p(ln[1]) if ln else None, float(eps), _p(ln_out), _p(mean), _p(rstd), _p(dact), _p(bits), _p(seg_order), _stream()),
        "bl_segment_max_fwd")
    res = (out, arg, ln_out, mean, rstd)
    if want_dact:
        res += (dact,)
    if want_bits:
        res += (bits,)
    return res


def segment_max_bwd(g_out, arg, x, seg_of, act=ACT_NONE, out=None):
    nitems, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    _check(
        load_library().bl_segment_max_bwd(_f32(g_out).data_ptr(), _i32(arg).data_ptr(), x.data_ptr(), x.stride(0),
                                          _i32(seg_of).data_ptr(), int(nitems), int(D), int(act), out.data_ptr(), _stream()),
        "bl_segment_max_bwd")
    return out


def layernorm_bwd(g_y, x, mean, rstd, gamma, g_gamma, g_beta, post_scale=None, want="f32"):
    """want: "f32" -> g_x; "packed" -> bf16x3-packed g_x only (int16 [n, 3 D]); "both" -> (g_x, packed)."""
    n, D = x.shape
    g_x = torch.empty_like(x) if want != "packed" else None
    g_xp = torch.empty((n, 3 * D), dtype=torch.int16, device=x.device) if want != "f32" else None
    _check(
        load_library().bl_layernorm_bwd(_f32(g_y).data_ptr(), _f32(x).data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        _f32(gamma).data_ptr(), int(n), int(D), _p(g_x), g_gamma.data_ptr(),
                                        g_beta.data_ptr(), _p(post_scale), _p(g_xp), _stream()),
        "bl_layernorm_bwd")
    return g_x if want == "f32" else (g_xp if want == "packed" else (g_x, g_xp))


def act_bwd(g_y, y, act, drop: Dropout = NO_DROPOUT, g_bias=None):
    n, N = y.shape
    g_z = torch.empty_like(y)
    _check(
        load_library().bl_act_bwd(_f32(g_y).data_ptr(), _f32(y).data_ptr(), int(n), int(N), y.stride(0), int(act), drop.c(),
                                  g_z.data_ptr(), _p(g_bias), _stream()),
        "bl_act_bwd")
    return g_z


def scatter_add_rows(src, col_off, width, idx, out):
    R = src.shape[0]
    _check(
        load_library().bl_scatter_add_rows(_f32(src).data_ptr(), src.stride(0), int(col_off), int(width), _i32(idx).data_ptr(),
                                           int(R), _f32(out).data_ptr(), out.stride(0), _stream()),
        "bl_scatter_add_rows")
    return out


# ------------------------------------------------------------------------------------------------
# autograd wrappers
def _take_saved(ctx):
    """What a Function's forward kept in `ctx.saved`, handed over ONCE: backward drops the references at once (activations are
    freed as the backward pass proceeds), so a second backward through the same graph has nothing to read."""
    saved = ctx.saved
    if saved is None:
        raise RuntimeError("hip_ops: this graph's buffers were freed by its first backward pass; retain_graph=True / a second "
                           "backward through the same forward is not supported by the hip_ops Functions")
    ctx.saved = None
    return saved


class GraphIndex(NamedTuple):
    """Device-side index arrays of one minibatch (buglab.data.collate.to_device)."""

    msg_src: torch.Tensor
    msg_tgt: torch.Tensor
    type_ptr: torch.Tensor
    tgt_ptr: torch.Tensor
    tgt_msgs: torch.Tensor
    src_ptr: torch.Tensor
    src_msgs: torch.Tensor
    num_nodes: int
    num_messages: int
    num_types: int
    node_order: Optional[torch.Tensor] = None  # processing order of the per-node kernels: high-degree nodes first
    num_hubs: int = -1  # leading entries of node_order that are hubs (-1: unknown, the kernels look at the first 4096)


POOLINGS = ("max", "sum", "mean")  # BL_POOL_MAX / _SUM / _MEAN


class _EmbedSubtokenMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, lens, drop: Dropout, tok_csr, before_pool: bool, comb: int):
        _f32(table, "embedding table")
        N, S = ids.shape
        V, H = table.shape
        out = torch.empty((N, H), dtype=torch.float32, device=table.device)
        argsub = torch.empty((N, H), dtype=torch.int8, device=table.device) if comb == 0 else None
        _check(
            load_library().bl_embed_subtoken_pool_fwd(table.data_ptr(), V, H, _i32(ids).data_ptr(), _i32(lens).data_ptr(), N, S, comb,
                                                      drop.c(), int(bool(before_pool)), out.data_ptr(), out.stride(0), _p(argsub), _stream()),
            "bl_embed_subtoken_pool_fwd")
        ctx.saved = (table, ids, lens, argsub, drop, V, H, tok_csr, int(bool(before_pool)), comb)
        return out

    @staticmethod
    def backward(ctx, g_out):
        table, ids, lens, argsub, drop, V, H, tok_csr, before_pool, comb = _take_saved(ctx)
        g_out = g_out.contiguous()
        N, S = ids.shape
        direct = _direct_small(table)
        g_table = direct if direct is not None else torch.zeros((V, H), dtype=torch.float32, device=g_out.device)
        if tok_csr is not None:
            occ, chunk_ptr, chunk_tok = tok_csr
            _check(
                load_library().bl_embed_subtoken_pool_bwd_sorted(g_out.data_ptr(), g_out.stride(0), _i32(occ).data_ptr(),
                                                                 _i32(chunk_ptr).data_ptr(), _i32(chunk_tok).data_ptr(),
                                                                 int(chunk_tok.shape[0]), _i32(lens).data_ptr(), _p(argsub), S, H, comb, drop.c(),
                                                                 before_pool, g_table.data_ptr(), _stream()),
                "bl_embed_subtoken_pool_bwd_sorted")
        else:
            _check(
                load_library().bl_embed_subtoken_pool_bwd(g_out.data_ptr(), g_out.stride(0), ids.data_ptr(), _i32(lens).data_ptr(), _p(argsub),
                                                          N, S, H, V, comb, drop.c(), before_pool, g_table.data_ptr(), _stream()),
                "bl_embed_subtoken_pool_bwd")
        return (None if direct is not None else g_table), None, None, None, None, None, None


def embed_subtoken_max(table, ids, lens, drop: Dropout = NO_DROPOUT, tok_csr=None, dropout_before_pooling: bool = False,
                       combination: str = "max"):
    """tok_csr = (occ, chunk_ptr, chunk_tok) from the collator (token-sorted subtoken occurrences): backward
    then sums per token in registers instead of issuing one atomic per (node, channel).
    dropout_before_pooling: dropout on the embedded subtokens (then max) instead of on the pooled rows.
    combination: "max" (the registry's default, modelregistry.py:65-66), "sum" or "mean" over the subtokens."""
    return _EmbedSubtokenMax.apply(table, ids, lens, drop, tok_csr, bool(dropout_before_pooling), POOLINGS.index(combination))


class _MpLayer(torch.autograd.Function):
    """One MlpMessagePassingLayer: message GEMM -> segmented max(+GELU) + LayerNorm -> dense+tanh+dropout.

    Saved for backward: only per-NODE arrays -- argmax [N, Dm] int32, the message activation's
    derivative at the winner `dact` [N, Dm], the aggregate, LayerNorm statistics/output and the layer
    output.  The [E, Dm] messages are dropped right after the segmented max: backward re-creates the
    (80 % zero) message gradient on the fly inside the two GEMMs' operand loads
    (bl_gemm_rows_routed / bl_gemm_wgrad_routed) from the node gradient and the winner table."""

    @staticmethod
    def forward(ctx, h, W, ln_g, ln_b, Wd, bd, g: GraphIndex, msg_act: int, drop: Dropout):
        _f32(h, "node states")
        N, Din = h.shape
        T, K2, Dm = W.shape
        Dout = Wd.shape[1]
        E = g.num_messages
        assert K2 == 2 * Din and T == g.num_types and N == g.num_nodes
        if x6_ok(Din, Dm):
            hp = pack_bf16x3(h)                   # [N, 3*Din]
            wtp = pack_weights_x6(_f32(W, "W"), True)
            pre = gemm_rows_x6([(hp, g.msg_src, Din), (hp, g.msg_tgt, Din)], wtp, E, Dm, group_ptr=g.type_ptr, G=T)
            del wtp
            if not WGRAD_X6:
                hp = None
        else:
            hp = None
            pre = gemm_rows([(h, g.msg_src), (h, g.msg_tgt)], _f32(W, "W"), E, Dm, b_group_stride=K2 * Dm, ldb=Dm,
                            group_ptr=g.type_ptr, G=T)
        use_bits = x6_ok(Din, Dm)
        res = segment_max(pre, g.tgt_ptr, g.tgt_msgs, N, act=msg_act, ln=(_f32(ln_g), _f32(ln_b)), want_dact=True, want_bits=use_bits,
                          seg_order=g.node_order)
        agg, arg, ln_out, mean, rstd, dact = res[:6]
        bits = res[6] if use_bits else None
        if WINNER_SINK is not None:
            WINNER_SINK.append(arg.clone())
        if use_bits and WGRAD_X6:
            arg = None  # the bf16x6 backward routes with the per-message bitmask only
        del pre, res
        if msg_act == ACT_NONE:
            dact = None  # derivative is identically 1
        out = gemm_rows([(ln_out, None)], _f32(Wd, "Wd"), N, Dout, bias=_f32(bd), act=ACT_TANH, drop=drop)
        # (the OUTPUT goes through save_for_backward: output -> grad_fn -> ctx -> output held as a plain attribute is a cycle
        # across the C++ boundary that nothing collects when no backward pass runs -- see _GatherLinear)
        ctx.save_for_backward(out)
        ctx.saved = (h, hp, W, ln_g, ln_b, Wd, bd, dact, arg, bits, agg, mean, rstd, ln_out, g, msg_act, drop)
        return out

    @staticmethod
    def backward(ctx, g_out):
        h, hp, W, ln_g, ln_b, Wd, bd, dact, arg, bits, agg, mean, rstd, ln_out, g, msg_act, drop = _take_saved(ctx)
        (out,) = ctx.saved_tensors
        N, Din = h.shape
        T, K2, Dm = W.shape
        Dout = Wd.shape[1]
        E = g.num_messages
        dev = h.device
        g_out = g_out.contiguous()
        # dense + tanh + dropout
        bd_direct, lng_direct, lnb_direct = _direct_small(bd), _direct_small(ln_g), _direct_small(ln_b)
        g_bd = bd_direct if bd_direct is not None else torch.zeros((Dout,), dtype=torch.float32, device=dev)
        g_z = act_bwd(g_out, out, ACT_TANH, drop, g_bd)
        Wd_direct, W_direct = _direct_grad_target(Wd), _direct_grad_target(W)
        g_Wd = Wd_direct if Wd_direct is not None else torch.zeros_like(Wd)
        g_W = W_direct if W_direct is not None else torch.zeros_like(W)
        side1 = _on_side_stream(dev)
        with side1:
            gemm_wgrad([(ln_out, None)], g_z, N, Dout, g_Wd)
        g_ln = gemm_rows([(g_z, None)], Wd, N, Dm, b_is_nk=True, ldb=Dout)
        # LayerNorm
        g_lng = lng_direct if lng_direct is not None else torch.zeros((Dm,), dtype=torch.float32, device=dev)
        g_lnb = lnb_direct if lnb_direct is not None else torch.zeros((Dm,), dtype=torch.float32, device=dev)
        # LayerNorm (+ the message activation's derivative at each winner): d loss / d (winning pre-activation) per node;
        # the bf16x6 GEMMs take it packed, straight from the LayerNorm kernel
        use_x6 = x6_ok(Din, Dm) and bits is not None
        if use_x6 and hp is not None:
            gq, gqp = None, layernorm_bwd(g_ln, agg, mean, rstd, ln_g, g_lng, g_lnb, post_scale=dact, want="packed")
        elif use_x6:
            gq, gqp = layernorm_bwd(g_ln, agg, mean, rstd, ln_g, g_lng, g_lnb, post_scale=dact, want="both")
        else:
            gq, gqp = layernorm_bwd(g_ln, agg, mean, rstd, ln_g, g_lng, g_lnb, post_scale=dact), None
        if bd_direct is not None:
            g_bd = None
        if lng_direct is not None:
            g_lng = None
        if lnb_direct is not None:
            g_lnb = None
        # per-edge-type weights; message m's gradient row = gq[tgt(m)] masked to the channels m won
        pair = _timed("mp_bwd_gemm_pair(wgrad||dgrad+node-sums)", 2.0 * (2.0 * E * K2 * Dm), span=True)
        pair.__enter__()
        side2 = _on_side_stream(dev)
        with side2:
            if hp is not None:
                gemm_wgrad_routed_x6([(hp, g.msg_src, Din), (hp, g.msg_tgt, Din)], gqp, g.msg_tgt, bits, E, Dm, g_W,
                                     gw_group_stride=K2 * Dm, group_ptr=g.type_ptr, G=T)
            else:
                gemm_wgrad_routed([(h, g.msg_src), (h, g.msg_tgt)], gq, g.msg_tgt, arg, E, Dm, g_W, gw_group_stride=K2 * Dm,
                                  group_ptr=g.type_ptr, G=T)
        # node states: per-message input gradients, then segmented sums over the src / tgt CSRs
        if gqp is not None:
            # d a = G . W_t^T: B_g = W_t itself read as [n = 2*Din, k = Dm]
            g_a = gemm_rows_x6([(gqp, g.msg_tgt, Dm)], pack_weights_x6(W, False), E, K2, group_ptr=g.type_ptr, G=T,
                               win_bits=bits, kind="gemm_rows_nk_routed_x6")
        else:
            g_a = gemm_rows_routed(gq, g.msg_tgt, arg, W, E, K2, b_group_stride=K2 * Dm, ldb=Dm, group_ptr=g.type_ptr, G=T)
        g_h = torch.empty((N, Din), dtype=torch.float32, device=dev)
        _check(
            load_library().bl_mp_scatter_grad(g_a.data_ptr(), g_a.stride(0), g.src_ptr.data_ptr(), g.src_msgs.data_ptr(),
                                              g.tgt_ptr.data_ptr(), g.tgt_msgs.data_ptr(), N, Din, 0, g_h.data_ptr(),
                                              g_h.stride(0), _p(g.node_order), _stream()),
            "bl_mp_scatter_grad")
        if W_direct is not None and Wd_direct is not None:
            # gradients land in param.grad behind the main chain; joined by join_side_stream()
            _streams.mark_free_running()
            side2.detach(h, gq, arg, bits, hp, gqp)
            side1.detach(ln_out, g_z)
            pair.__exit__(None, None, None)
            return g_h, None, g_lng, g_lnb, None, g_bd, None, None, None
        side2.join()
        side1.join()
        pair.__exit__(None, None, None)
        if W_direct is not None:
            g_W = None
        if Wd_direct is not None:
            g_Wd = None
        return g_h, g_W, g_lng, g_lnb, g_Wd, g_bd, None, None, None



# ------------------------------------------------------------------------------------------------
# One C call per message-passing layer and direction (bl_mp_layer_fwd / bl_mp_layer_bwd).
FUSED_LAYER = os.environ.get("BL_FUSED_LAYER", "1") != "0"
_weights_epoch = 0          # bumped by whoever changes parameters behind autograd's back (FlatAdam's kernel)


def set_deterministic(on: bool = True) -> None:
    """Bit-reproducible gradients (ordered flushes instead of free-running atomics; slower).  BL_DETERMINISTIC=1 in the
    environment does the same for this process AND the loader processes (the collator must keep every token's
    occurrences in one chunk); this call only reaches the collators of this process."""
    load_library().bl_set_deterministic(1 if on else 0)
    os.environ["BL_DETERMINISTIC"] = "1" if on else "0"


_MSG_GEMM_MODES = ("bf16x6", "f16x3", "f16x1")


def set_msg_gemm_mode(mode: str) -> str:
    """'f16x3' (default), 'bf16x6' or 'f16x1': the operand split of the message GEMMs inside the fused layer calls
    (bl_set_msg_gemm_mode).  'f16x1' is the reduced-precision mode of `train.py --amp` (reference train.py:8,106: autocast): the
    f16x3 images with the high-plane term only -- fp16 operands, fp32 accumulation, fp32 results; outside the 1e-4 parity bound
    by construction and never the benchmarked headline.  Returns the previous mode.  Not to be switched between a forward pass
    and its backward pass."""
    if mode not in _MSG_GEMM_MODES:
        raise ValueError(f"mode must be one of {_MSG_GEMM_MODES}")
    prev = load_library().bl_set_msg_gemm_mode(_MSG_GEMM_MODES.index(mode))
    return _MSG_GEMM_MODES[prev]


def msg_gemm_mode() -> str:
    return _MSG_GEMM_MODES[load_library().bl_get_msg_gemm_mode()]


def set_wgrad_tile(rows: int) -> int:
    """256 (default): wide weight-gradient tile where it applies; 128: the 128 x 128 tile everywhere.  -> previous value."""
    return int(load_library().bl_set_wgrad_tile(int(rows)))


def set_wgrad_kchunk_cap(rows: int) -> int:
    """Most rows a workgroup of the bf16x6 weight-gradient GEMMs reduces per output-tile flush.  -> previous value."""
    return int(load_library().bl_set_wgrad_kchunk_cap(int(rows)))


def set_fused_node_bwd(on: bool) -> bool:
    """A/B switch: the node update's backward chain of the fused layer call as one kernel (default) or three.  -> previous."""
    return bool(load_library().bl_set_fused_node_bwd(1 if on else 0))


def node_update_bwd(g_out, h_out, drop: "Dropout", wd_packed_bwd, agg, mean, rstd, ln_g, dact, g_bias, g_ln_g, g_ln_b, want_f32=True):
    """bl_node_update_bwd (csrc/bl_node_bwd.hip) -> (packed g_z [N, 3 Dout] int16, gq fp32 [N, Dm] or None, packed gq [N, 3 Dm])."""
    N, Dout = g_out.shape
    Dm = agg.shape[1]
    dev = g_out.device
    gz = torch.empty((N, 3 * Dout), dtype=torch.int16, device=dev)
    gq = torch.empty((N, Dm), dtype=torch.float32, device=dev) if want_f32 else None
    gqp = torch.empty((N, 3 * Dm), dtype=torch.int16, device=dev)
    _check(load_library().bl_node_update_bwd(_f32(g_out).data_ptr(), _f32(h_out).data_ptr(), N, Dout, drop.c(), wd_packed_bwd.data_ptr(),
                                             _f32(agg).data_ptr(), mean.data_ptr(), rstd.data_ptr(), ln_g.data_ptr(), _p(dact), Dm,
                                             gz.data_ptr(), _p(g_bias), _p(gq), gqp.data_ptr(), g_ln_g.data_ptr(), g_ln_b.data_ptr(),
                                             _stream()), "bl_node_update_bwd")
    return gz, gq, gqp


def deterministic() -> bool:
    return bool(load_library().bl_get_deterministic())


def invalidate_weight_packs():
    """Parameters were updated in place by a kernel (no `_version` bump): packed copies are stale."""
    global _weights_epoch
    _weights_epoch += 1


# The dense node update (LayerNorm -> Linear -> tanh -> Dropout) as bf16x6 GEMMs too (forward, input gradient, weight
# gradient); BL_DENSE_X6=0: exact-fp32 MFMA GEMMs.
DENSE_X6 = os.environ.get("BL_DENSE_X6", "1") != "0"


def _as_groups(w: torch.Tensor) -> torch.Tensor:
    return w if w.dim() == 3 else w.unsqueeze(0)


# Operand copies of the weights (bf16x3-packed tiled forms for the bf16x6 GEMMs, fp32 transposes for the vector input
# gradient).  They are functions of the parameter values, so a training step re-makes all of them once after the optimiser
# step -- in ONE launch (bl_pack_weights_multi) over a table of every copy any layer has asked for so far, instead of one
# launch per layer and form.  Validity = (parameter object, its autograd version, the epoch bumped by whoever writes
# parameters behind autograd's back, its storage address).
# ..w: the wide row GEMM's image (bl_pack_weights_x6w); ..h: the f16x3 image (bl_pack_weights_h3, scale BL_H3_W_SCALE)
_KIND = {"nk": 0, "kn": 1, "t": 2, "nkw": 3, "knw": 4, "nkh": 5, "knh": 6}


class _WeightCopies:
    def __init__(self):
        self.entries = {}   # id(W) -> {"ref", "ptr", "shape", "forms": {name: tensor}, "version", "epoch"}
        self.plan = None    # (device job table, njobs, total_blocks, [entries in table order])

    def _fresh(self, ent, W) -> bool:
        return ent["version"] == W._version and ent["epoch"] == _weights_epoch

    def get(self, W: torch.Tensor, names):
        import weakref

        ent = self.entries.get(id(W))
        if ent is not None and (ent["ref"]() is not W or ent["ptr"] != W.data_ptr() or ent["shape"] != tuple(W.shape)):
            ent = None
        if ent is None:
            if len(self.entries) > 256:
                self.entries = {k: v for k, v in self.entries.items() if v["ref"]() is not None}
            ent = {"ref": weakref.ref(W), "ptr": W.data_ptr(), "shape": tuple(W.shape), "forms": {}, "version": -1, "epoch": -1}
            self.entries[id(W)] = ent
            self.plan = None
        G, K, N = _as_groups(W).shape  # the parameter is [G][K][N] (forward form: C = A . W[g])
        for nm in names:
            if nm not in ent["forms"]:
                if nm == "t":
                    ent["forms"][nm] = torch.empty((G, N, K), dtype=torch.float32, device=W.device)
                else:  # "kn": C = A . W (K x N) ; "nk": C = G . W^T, i.e. bl_pack_weights_x6 of [G][N'][K'] with N' = K, K' = N
                    n_out, k_in = (N, K) if nm.startswith("kn") else (K, N)
                    per = (int(load_library().bl_packed_weight_elems_x6w(1, k_in, n_out)) if nm.endswith("w")
                           else int(load_library().bl_packed_weight_elems_h3(1, k_in, n_out)) if nm.endswith("h")
                           else ((n_out + 127) // 128) * (k_in // 32) * 12288)
                    ent["forms"][nm] = torch.empty((G, per), dtype=torch.int16, device=W.device)
                ent["version"] = -1  # (a new form has to be filled)
                self.plan = None
        if not self._fresh(ent, W):
            self.refresh(W.device)
        return [ent["forms"][nm] for nm in names]

    def refresh(self, device) -> None:
        """Re-make every registered copy on `device` whose parameter changed: one launch."""
        lib = load_library()
        # (strong references for the duration: a parameter that is only kept alive by a reference cycle can be collected by
        # the cyclic GC at any allocation below)
        alive = [(e, e["ref"]()) for e in self.entries.values()]
        alive = [(e, W) for e, W in alive if W is not None and W.device == device]
        live = [e for e, _ in alive]
        if self.plan is None or self.plan[4] != device or len(self.plan[3]) != len(live) or any(a is not b for a, b in zip(self.plan[3], live)):
            jobs, blocks = [], 0
            for e, W in alive:
                G, K, N = _as_groups(W).shape
                for nm, out in e["forms"].items():
                    j = bl_pack_job_t()
                    j.w, j.out, j.kind = W.data_ptr(), out.data_ptr(), _KIND[nm]
                    # kind 1 (kn): w [G][K][N]; kind 0 (nk): bl_pack_weights_x6(w_is_kn = 0) reads w as [G][N'][K'] = [G][K][N]
                    # with output columns N' = K and contraction K' = N; kind 2: transpose of [G][K][N]
                    j.G, j.K, j.N = (G, K, N) if not nm.startswith("nk") else (G, N, K)
                    j.first_block = blocks
                    blocks += int(lib.bl_pack_job_blocks(j.kind, j.G, j.K, j.N))
                    jobs.append(j)
            raw = b"".join(bytes(j) for j in jobs)
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device) if jobs else None
            self.plan = (table, len(jobs), blocks, live, device)
        table, njobs, blocks, _, _ = self.plan
        if njobs:
            _check(lib.bl_pack_weights_multi(table.data_ptr(), njobs, blocks, _stream()), "bl_pack_weights_multi")
        for e, W in alive:
            e["version"], e["epoch"] = W._version, _weights_epoch


_weight_copies = _WeightCopies()


def _packed_layer_weights(W: torch.Tensor, need_bwd: bool):
    """bf16x3-packed, tiled copies of a layer's weights: the forward form (C = A . W[t]) and, when asked for, the form of the
    input-gradient GEMM (C = G . W[t]^T).  A 2-D weight (the dense node update's Wd [Dm, Dout]) is one group."""
    got = _weight_copies.get(W, ("kn", "nk") if need_bwd else ("kn",))
    return got[0], (got[1] if need_bwd else None)


def _packed_message_weights(W: torch.Tensor, Din: int, need_bwd: bool):
    """The per-type message weights W [T, 2 Din, Dm] in the images the fused layer calls expect (bl_mp_layer_weight_image): the wide
    row GEMM's image where that kernel takes the shape, the tiled one otherwise."""
    lib = load_library()
    Dm = W.shape[2]
    suffix = ("", "w", "h")  # bl_mp_layer_weight_image: 0 = 128 x 128 bf16x6 image, 1 = wide bf16x6 image, 2 = f16x3 image
    fwd = "kn" + suffix[int(lib.bl_mp_layer_weight_image(int(Din), int(Dm), 0))]
    if not need_bwd:
        return _weight_copies.get(W, (fwd,))[0], None
    bwd = "nk" + suffix[int(lib.bl_mp_layer_weight_image(int(Din), int(Dm), 1))]
    got = _weight_copies.get(W, (fwd, bwd))
    return got[0], got[1]


# The routed input gradient of a message-passing layer from the NON-ZEROS of the message gradient, on the vector units, node
# sums fused in (csrc/bl_routed_dgrad.hip), instead of the matrix-core GEMM over all E x Dm entries + bl_mp_scatter_grad.
# BL_DGRAD_VEC=0: matrix cores.  Needs W transposed ([T, Dm, 2 Din]); cached per parameter value like the packed forms.
# Default (BL_DGRAD_VEC unset): vector units while the message GEMMs run as bf16x6 (0.404 vs 0.550 ms per hidden-128 layer), matrix
# cores when they run as f16x3 -- the routed f16x3 GEMM + segmented sums cost the same exclusive time as the vector kernel + its
# sums (4.24 vs 4.33 ms per c2 step) and overlap better with the free-running weight gradients (the vector kernel holds a CU's
# whole LDS with one 1024-thread workgroup): 13.12 vs 13.68 ms per step (profiles/r06h_bench*.json).
DGRAD_VEC = {"1": True, "0": False}.get(os.environ.get("BL_DGRAD_VEC", ""), None)


def _use_vector_dgrad(lib, E: int, Dm: int, K2: int) -> bool:
    want = DGRAD_VEC if DGRAD_VEC is not None else not lib.bl_get_msg_gemm_mode()
    return bool(want) and E > 0 and bool(lib.bl_routed_dgrad_vec_ok(Dm, K2))


def _transposed_layer_weights(W: torch.Tensor) -> torch.Tensor:
    return _weight_copies.get(W, ("t",))[0]


AGGREGATIONS = ("max", "sum", "mean")  # BL_AGG_MAX / _SUM / _MEAN (ptgnn's message_aggregation_function values)


def _layer_desc(g: "GraphIndex", W, ln_g, ln_b, Wd, bd, Din, msg_act, drop: Dropout, agg: int = 0) -> bl_mp_layer_t:
    L = bl_mp_layer_t()
    L.aggregation = int(agg)
    L.N, L.E, L.T, L.Din, L.Dm, L.Dout = g.num_nodes, g.num_messages, W.shape[0], Din, W.shape[2], Wd.shape[1]
    L.msg_src, L.msg_tgt, L.type_ptr = g.msg_src.data_ptr(), g.msg_tgt.data_ptr(), g.type_ptr.data_ptr()
    L.tgt_ptr, L.tgt_msgs, L.src_ptr, L.src_msgs = g.tgt_ptr.data_ptr(), g.tgt_msgs.data_ptr(), g.src_ptr.data_ptr(), g.src_msgs.data_ptr()
    L.node_order = _p(g.node_order)
    L.num_hub_slots = int(g.num_hubs)
    L.W, L.ln_g, L.ln_b, L.Wd, L.bd = W.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), Wd.data_ptr(), bd.data_ptr()
    L.msg_act, L.ln_eps, L.drop = int(msg_act), 1e-5, drop.c()
    return L


INFERENCE_MODE = True  # forward-only form of the fused layer call when no input needs a gradient (A/B switch for tests)


class _MpLayerFused(torch.autograd.Function):
    """One MlpMessagePassingLayer = one C call forward, one backward.  The layer input is `h_lo` alone or the
    virtual concatenation [h_lo ; h_hi] of a ConcatResidual layer (never materialised).  What forward keeps for
    backward is one opaque byte blob (packed input, routing bitmask, LayerNorm state; layout in csrc/bl_mp_layer.hip)."""

    @staticmethod
    def forward(ctx, h_lo, h_hi, W, ln_g, ln_b, Wd, bd, g: GraphIndex, msg_act: int, drop: Dropout, agg: int = 0):
        _f32(h_lo, "node states")
        lib = load_library()
        N = h_lo.shape[0]
        Din = h_lo.shape[1] + (h_hi.shape[1] if h_hi is not None else 0)
        T, K2, Dm = W.shape
        Dout = Wd.shape[1]
        E = g.num_messages
        assert K2 == 2 * Din and T == g.num_types and N == g.num_nodes
        for t, nm in ((W, "W"), (ln_g, "ln_g"), (ln_b, "ln_b"), (Wd, "Wd"), (bd, "bd")):
            _f32(t, nm)
        # (grad mode is always off inside Function.forward: whether a backward pass will follow is in needs_input_grad)
        need_bwd = any(ctx.needs_input_grad[:7])
        # which form of W the input gradient will read: its fp32 transpose (vector-unit path) or the packed C = G . W^T form
        use_vec = agg == 0 and _use_vector_dgrad(lib, E, Dm, K2)  # (sum / mean: no routing bits -> matrix-core input gradient)
        wkn, wnk = _packed_message_weights(W, Din, need_bwd and not use_vec)
        wt = _transposed_layer_weights(W) if (need_bwd and use_vec) else None
        dev = h_lo.device
        L = _layer_desc(g, W, ln_g, ln_b, Wd, bd, Din, msg_act, drop, agg)
        dense_x6 = DENSE_X6 and Dm % 32 == 0 and Dout % 32 == 0
        wd_kn = wd_nk = None
        if dense_x6:
            wd_kn, wd_nk = _packed_layer_weights(Wd, need_bwd)
            L.Wd_packed = wd_kn.data_ptr()
        # no backward pass will follow (predict / evaluate under no_grad): nothing is saved, the call skips every store that
        # only a backward pass reads (routing bitmask, activation derivative, aggregate, LayerNorm statistics)
        infer = not need_bwd and INFERENCE_MODE
        # ("mean" without an activation keeps the derivative array all the same: it carries the 1 / in-degree)
        saved_act = ACT_GELU_AGG if (agg == 2 and msg_act == ACT_NONE) else msg_act
        saved = None if infer else torch.empty((lib.bl_mp_layer_saved_bytes(N, E, Din, Dm, saved_act),), dtype=torch.uint8, device=dev)
        ws = torch.empty((lib.bl_mp_layer_workspace_bytes(N, E, Din, Dm, Dout, 3 if infer else 0),), dtype=torch.uint8, device=dev)
        out = torch.empty((N, Dout), dtype=torch.float32, device=dev)
        winner = torch.empty((N, Dm), dtype=torch.int32, device=dev) if (WINNER_SINK is not None and agg == 0) else None
        _check(lib.bl_mp_layer_fwd(ctypes.byref(L), h_lo.data_ptr(), h_lo.stride(0), h_lo.shape[1], _p(h_hi),
                                   h_hi.stride(0) if h_hi is not None else 0, wkn.data_ptr(), out.data_ptr(), _p(winner),
                                   _p(saved), ws.data_ptr() if (E > 0 or infer) else None, _stream()), "bl_mp_layer_fwd")
        if winner is not None:
            WINNER_SINK.append(winner)
        if need_bwd:
            _note_use((W, ln_g, ln_b, Wd, bd))
        ctx.save_for_backward(out)  # (an output: never as a plain ctx attribute, see _MpLayer)
        ctx.saved = (h_lo.shape[1], h_hi.shape[1] if h_hi is not None else 0, W, ln_g, ln_b, Wd, bd, g, msg_act, drop, saved, wnk,
                     dense_x6, wd_kn, wd_nk, wt, agg)
        return out

    @staticmethod
    def backward(ctx, g_out):
        w_lo, w_hi, W, ln_g, ln_b, Wd, bd, g, msg_act, drop, saved, wnk, dense_x6, wd_kn, wd_nk, wt, agg = _take_saved(ctx)
        (out,) = ctx.saved_tensors
        lib = load_library()
        N, E = g.num_nodes, g.num_messages
        T, K2, Dm = W.shape
        Din, Dout = w_lo + w_hi, Wd.shape[1]
        dev = out.device
        g_out = g_out.contiguous()
        use_vec = agg == 0 and _use_vector_dgrad(lib, E, Dm, 2 * Din)
        if use_vec and wt is None:
            wt = _transposed_layer_weights(W)
        if wnk is None and not use_vec:  # forward ran without grad mode knowing a backward would follow
            wnk = _packed_message_weights(W, Din, True)[1]
        direct = [_direct_small(bd), _direct_small(ln_g), _direct_small(ln_b), _direct_grad_target(Wd), _direct_grad_target(W)]
        tgt = [d if d is not None else torch.zeros_like(p) for d, p in zip(direct, (bd, ln_g, ln_b, Wd, W))]
        g_bd, g_lng, g_lnb, g_Wd, g_W = tgt
        L = _layer_desc(g, W, ln_g, ln_b, Wd, bd, Din, msg_act, drop, agg)
        if dense_x6:  # (forward kept the LayerNorm output in packed form: backward must take the same path)
            if wd_nk is None:
                wd_nk = pack_weights_x6(_as_groups(Wd.detach()), False)
            L.Wd_packed, L.Wd_packed_bwd = wd_kn.data_ptr(), wd_nk.data_ptr()
        ws_mode = 1
        if use_vec:
            L.Wt = wt.data_ptr()
            if not lib.bl_get_deterministic():
                ws_mode = 2  # node sums fused into the input-gradient kernel: no [E, 2 Din] scratch
        ws = torch.empty((lib.bl_mp_layer_workspace_bytes(N, E, Din, Dm, Dout, ws_mode),), dtype=torch.uint8, device=dev)
        g_lo = torch.empty((N, w_lo), dtype=torch.float32, device=dev)
        g_hi = torch.empty((N, w_hi), dtype=torch.float32, device=dev) if w_hi else None
        side = _streams.side_stream_for_current_device()
        free_running = side is not None and direct[3] is not None and direct[4] is not None
        _check(lib.bl_mp_layer_bwd(ctypes.byref(L), out.data_ptr(), g_out.data_ptr(), _p(wnk), saved.data_ptr(), ws.data_ptr(),
                                   g_lo.data_ptr(), g_lo.stride(0), w_lo, _p(g_hi), g_hi.stride(0) if g_hi is not None else 0,
                                   g_W.data_ptr(), g_lng.data_ptr(), g_lnb.data_ptr(), g_Wd.data_ptr(), g_bd.data_ptr(), _stream(),
                                   side.cuda_stream if side is not None else None, 0 if free_running else 1), "bl_mp_layer_bwd")
        if free_running:
            # the two weight-gradient GEMMs keep running behind the main chain (joined by join_side_stream()):
            # what they read must not be recycled by the allocator before they are done -- held until the join
            _streams.mark_free_running(saved, ws)
        ret = [None if d is not None else t for d, t in zip(direct, tgt)]
        if all(d is not None for d in direct):  # (gradients returned through autograd are not in place yet)
            _notify_backward_launched((W, ln_g, ln_b, Wd, bd))
        return g_lo, g_hi, ret[4], ret[1], ret[2], ret[3], ret[0], None, None, None, None


# ---- "the backward of this layer has been launched" notifications (data-parallel gradient buckets, runtime/optim.py) ----
GRAD_READY_CALLBACK = None
_pending_uses = {}  # id(param) -> forward uses whose backward has not been launched yet (weight sharing)


def set_grad_ready_callback(fn) -> None:
    """fn(list of parameters) is called from a message-passing layer's backward once every kernel that adds into those
    parameters' gradients has been LAUNCHED (on the training stream or the side stream); None switches it off."""
    global GRAD_READY_CALLBACK
    GRAD_READY_CALLBACK = fn
    _pending_uses.clear()


def _note_use(params) -> None:
    if GRAD_READY_CALLBACK is not None:
        for p in params:
            _pending_uses[id(p)] = _pending_uses.get(id(p), 0) + 1


def _notify_backward_launched(params) -> None:
    if GRAD_READY_CALLBACK is None:
        return
    done = []
    for p in params:
        n = _pending_uses.get(id(p), 1) - 1
        if n <= 0:
            _pending_uses.pop(id(p), None)
            done.append(p)
        else:
            _pending_uses[id(p)] = n
    if done:
        GRAD_READY_CALLBACK(done)


def fused_layer_ok(Din: int, Dm: int) -> bool:
    return FUSED_LAYER and WGRAD_X6 and x6_ok(Din, Dm) and Dm <= 512


class _GatedMpLayer(torch.autograd.Function):
    """One GatedMessagePassingLayer (GGNN): m_e = h[src] @ W[type] -> segmented max -> GRU cell(+dropout).
    Backward keeps per-node arrays only (winner table + GRU gate pre-activations)."""

    @staticmethod
    def forward(ctx, h, W, Wi, bi, Wh, bh, g: GraphIndex, drop: Dropout):
        _f32(h, "node states")
        N, D = h.shape
        T, Din, Dm = W.shape
        E = g.num_messages
        assert Din == D and Wi.shape == (Dm, 3 * D) and Wh.shape == (D, 3 * D)
        msgs = gemm_rows([(h, g.msg_src)], _f32(W, "W"), E, Dm, b_group_stride=D * Dm, ldb=Dm, group_ptr=g.type_ptr, G=T)
        agg, arg, _, _, _ = segment_max(msgs, g.tgt_ptr, g.tgt_msgs, N, seg_order=g.node_order)
        del msgs
        gi = gemm_rows([(agg, None)], _f32(Wi), N, 3 * D, bias=_f32(bi))
        gh = gemm_rows([(h, None)], _f32(Wh), N, 3 * D, bias=_f32(bh))
        out = torch.empty((N, D), dtype=torch.float32, device=h.device)
        _check(load_library().bl_gru_cell_fwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), h.stride(0), N, D, drop.c(), out.data_ptr(), _stream()),
               "bl_gru_cell_fwd")
        ctx.saved = (h, W, Wi, Wh, agg, arg, gi, gh, g, drop)
        return out

    @staticmethod
    def backward(ctx, g_out):
        h, W, Wi, Wh, agg, arg, gi, gh, g, drop = _take_saved(ctx)
        N, D = h.shape
        T, _, Dm = W.shape
        E = g.num_messages
        dev = h.device
        g_out = g_out.contiguous()
        g_gi = torch.empty_like(gi)
        g_gh = torch.empty_like(gh)
        g_h = torch.empty_like(h)
        _check(load_library().bl_gru_cell_bwd(g_out.data_ptr(), gi.data_ptr(), gh.data_ptr(), h.data_ptr(), h.stride(0), N, D, drop.c(),
                                              g_gi.data_ptr(), g_gh.data_ptr(), g_h.data_ptr(), _stream()), "bl_gru_cell_bwd")
        g_bi, g_bh = g_gi.sum(0), g_gh.sum(0)
        g_Wi, g_Wh, g_W = torch.zeros_like(Wi), torch.zeros_like(Wh), torch.zeros_like(W)
        side = _on_side_stream(dev)
        with side:
            gemm_wgrad([(agg, None)], g_gi, N, 3 * D, g_Wi)
            gemm_wgrad([(h, None)], g_gh, N, 3 * D, g_Wh)
        g_h += gemm_rows([(g_gh, None)], Wh, N, D, b_is_nk=True, ldb=3 * D)
        gq = gemm_rows([(g_gi, None)], Wi, N, Dm, b_is_nk=True, ldb=3 * D)  # d loss / d aggregate
        with side:
            gemm_wgrad_routed([(h, g.msg_src)], gq, g.msg_tgt, arg, E, Dm, g_W, gw_group_stride=D * Dm, group_ptr=g.type_ptr, G=T)
        g_a = gemm_rows_routed(gq, g.msg_tgt, arg, W, E, D, b_group_stride=D * Dm, ldb=Dm, group_ptr=g.type_ptr, G=T)
        _check(
            load_library().bl_mp_scatter_grad(g_a.data_ptr(), g_a.stride(0), g.src_ptr.data_ptr(), g.src_msgs.data_ptr(), None, None,
                                              N, D, 1, g_h.data_ptr(), g_h.stride(0), _p(g.node_order), _stream()),
            "bl_mp_scatter_grad")
        side.join()
        side.join()
        return g_h, g_W, g_Wi, g_bi, g_Wh, g_bh, None, None


def gated_mp_layer(h, W, Wi, bi, Wh, bh, graph: GraphIndex, drop: Dropout = NO_DROPOUT):
    return _GatedMpLayer.apply(h.contiguous(), W, Wi, bi, Wh, bh, graph, drop)


class _MpLayerFeat(torch.autograd.Function):
    """MlpMessagePassingLayer with edge features (`features_dimension` F > 0, reference gnnlayerdefs.py:13,22): the message
    input is [h_src ; h_tgt ; f_e] with f_e = edge_table[msg_feat[e]] read as a THIRD gathered source of the message GEMM --
    the [E, F] feature matrix is never materialised.  The three message GEMMs (forward, routed weight gradient, routed
    input gradient) run on the bf16x6 kernels with three packed sources when Din, Dm and F are multiples of 32 (the table
    is packed once per call like the node states); otherwise on the exact-fp32 kernels.  Kernel by kernel (non-default
    configuration), dense node update on the exact-fp32 GEMMs."""

    @staticmethod
    def forward(ctx, h, W, ln_g, ln_b, Wd, bd, table, msg_feat, g: GraphIndex, msg_act: int, drop: Dropout):
        _f32(h, "node states")
        N, Din = h.shape
        T, K3, Dm = W.shape
        F = table.shape[1]
        Dout = Wd.shape[1]
        E = g.num_messages
        assert K3 == 2 * Din + F and T == g.num_types and N == g.num_nodes and msg_feat.shape[0] == E
        use_x6 = x6_ok(Din, Dm, F) and WGRAD_X6
        hp = tp = bits = None
        if use_x6:
            hp, tp = pack_bf16x3(h), pack_bf16x3(_f32(table, "edge table"))
            pre = gemm_rows_x6([(hp, g.msg_src, Din), (hp, g.msg_tgt, Din), (tp, msg_feat, F)], _packed_layer_weights(W, False)[0], E, Dm,
                               group_ptr=g.type_ptr, G=T)
        else:
            src3 = [(h, g.msg_src), (h, g.msg_tgt), (_f32(table, "edge table"), msg_feat)]
            pre = gemm_rows(src3, _f32(W, "W"), E, Dm, b_group_stride=K3 * Dm, ldb=Dm, group_ptr=g.type_ptr, G=T)
        res = segment_max(pre, g.tgt_ptr, g.tgt_msgs, N, act=msg_act, ln=(_f32(ln_g), _f32(ln_b)), want_dact=True, want_bits=use_x6,
                          seg_order=g.node_order)
        agg, arg, ln_out, mean, rstd, dact = res[:6]
        if use_x6:
            bits = res[6]
        if WINNER_SINK is not None:
            WINNER_SINK.append(arg.clone())
        if use_x6:
            arg = None  # the bf16x6 backward routes with the per-message bitmask only
        del pre, res
        if msg_act == ACT_NONE:
            dact = None
        out = gemm_rows([(ln_out, None)], _f32(Wd, "Wd"), N, Dout, bias=_f32(bd), act=ACT_TANH, drop=drop)
        ctx.save_for_backward(out)  # (an output: never as a plain ctx attribute, see _MpLayer)
        ctx.saved = (h, hp, tp, bits, W, ln_g, Wd, table, msg_feat, dact, arg, agg, mean, rstd, ln_out, g, drop)
        return out

    @staticmethod
    def backward(ctx, g_out):
        h, hp, tp, bits, W, ln_g, Wd, table, msg_feat, dact, arg, agg, mean, rstd, ln_out, g, drop = _take_saved(ctx)
        (out,) = ctx.saved_tensors
        N, Din = h.shape
        T, K3, Dm = W.shape
        F, Dout, E, dev = table.shape[1], Wd.shape[1], g.num_messages, h.device
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        g_bd, g_Wd, g_W, g_lng, g_lnb, g_table = z(Dout), torch.zeros_like(Wd), torch.zeros_like(W), z(Dm), z(Dm), torch.zeros_like(table)
        g_z = act_bwd(g_out.contiguous(), out, ACT_TANH, drop, g_bd)
        gemm_wgrad([(ln_out, None)], g_z, N, Dout, g_Wd)
        g_ln = gemm_rows([(g_z, None)], Wd, N, Dm, b_is_nk=True, ldb=Dout)
        if hp is not None:
            # d loss / d (winning pre-activation) per node, packed for the bf16x6 GEMMs; message e's gradient row is
            # gq[tgt(e)] masked to the channels e won (the routing bitmask)
            gqp = layernorm_bwd(g_ln, agg, mean, rstd, ln_g, g_lng, g_lnb, post_scale=dact, want="packed")
            gemm_wgrad_routed_x6([(hp, g.msg_src, Din), (hp, g.msg_tgt, Din), (tp, msg_feat, F)], gqp, g.msg_tgt, bits, E, Dm, g_W,
                                 gw_group_stride=K3 * Dm, group_ptr=g.type_ptr, G=T)
            g_a = gemm_rows_x6([(gqp, g.msg_tgt, Dm)], _packed_layer_weights(W, True)[1], E, K3, group_ptr=g.type_ptr, G=T, win_bits=bits,
                               kind="gemm_rows_nk_routed_x6")  # [E, 2 Din + F]
        else:
            gq = layernorm_bwd(g_ln, agg, mean, rstd, ln_g, g_lng, g_lnb, post_scale=dact)
            src3 = [(h, g.msg_src), (h, g.msg_tgt), (table, msg_feat)]
            gemm_wgrad_routed(src3, gq, g.msg_tgt, arg, E, Dm, g_W, gw_group_stride=K3 * Dm, group_ptr=g.type_ptr, G=T)
            g_a = gemm_rows_routed(gq, g.msg_tgt, arg, W, E, K3, b_group_stride=K3 * Dm, ldb=Dm, group_ptr=g.type_ptr, G=T)  # [E, 2 Din + F]
        g_h = torch.empty((N, Din), dtype=torch.float32, device=dev)
        _check(load_library().bl_mp_scatter_grad(g_a.data_ptr(), g_a.stride(0), g.src_ptr.data_ptr(), g.src_msgs.data_ptr(), g.tgt_ptr.data_ptr(),
                                                 g.tgt_msgs.data_ptr(), N, Din, 0, g_h.data_ptr(), g_h.stride(0), _p(g.node_order), _stream()),
               "bl_mp_scatter_grad")
        if E > 0:
            scatter_add_rows(g_a, 2 * Din, F, msg_feat, g_table)  # the feature columns go back to the table rows they came from
        return g_h, g_W, g_lng, g_lnb, g_Wd, g_bd, g_table, None, None, None, None


def mp_layer_with_edge_features(h, W, ln_g, ln_b, Wd, bd, table, msg_feat, graph: GraphIndex, msg_act: str = "gelu_aggregated",
                                drop: Dropout = NO_DROPOUT):
    """mp_layer with [h_src ; h_tgt ; table[msg_feat]] as the message input (W: [T, 2 Din + F, Dm])."""
    if isinstance(h, (tuple, list)):
        h = torch.cat(list(h), dim=-1)
    return _MpLayerFeat.apply(h.contiguous(), W, ln_g, ln_b, Wd, bd, table, msg_feat, graph, _ACTS[msg_act], drop)


def mp_layer(h, W, ln_g, ln_b, Wd, bd, graph: GraphIndex, msg_act: str = "gelu_aggregated", drop: Dropout = NO_DROPOUT,
             aggregation: str = "max"):
    """h: the node states [N, Din], or a pair (stash, current) standing for their concatenation (ConcatResidual).
    aggregation: "max" (the reference's recipe, gnnlayerdefs.py:11,21) or ptgnn's "sum" / "mean" (one-call layer form only)."""
    pair = isinstance(h, (tuple, list))
    Din = sum(t.shape[1] for t in h) if pair else h.shape[1]
    agg = AGGREGATIONS.index(aggregation)
    if fused_layer_ok(Din, W.shape[2]) and (not pair or h[0].shape[1] % 32 == 0):
        lo, hi = (h[0].contiguous(), h[1].contiguous()) if pair else (h.contiguous(), None)
        return _MpLayerFused.apply(lo, hi, W, ln_g, ln_b, Wd, bd, graph, _ACTS[msg_act], drop, agg)
    if agg != 0:
        raise NotImplementedError("sum / mean message aggregation runs in the one-call layer form only (state and message widths multiples of "
                                  "32, message width <= 512, FUSED_LAYER on)")
    if pair:
        h = torch.cat(list(h), dim=-1)
    return _MpLayer.apply(h.contiguous(), W, ln_g, ln_b, Wd, bd, graph, _ACTS[msg_act], drop)


class _GatherRows(torch.autograd.Function):
    """x[idx] as a compact copy; backward = one zero-filled [N, H] buffer + one scatter-add."""

    @staticmethod
    def forward(ctx, x, idx):
        _f32(x, "x")
        R, H = idx.shape[0], x.shape[1]
        out = torch.empty((R, H), dtype=torch.float32, device=x.device)
        _check(load_library().bl_gather_rows(x.data_ptr(), x.stride(0), _i32(idx).data_ptr(), R, H, out.data_ptr(), out.stride(0), _stream()),
               "bl_gather_rows")
        ctx.saved = (idx, x.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        idx, shape = _take_saved(ctx)
        g_x = torch.zeros(shape, dtype=torch.float32, device=g_out.device)
        scatter_add_rows(g_out.contiguous(), 0, shape[1], idx, g_x)
        return g_x, None


def gather_rows(x, idx):
    return _GatherRows.apply(x, idx)


# Plain Linear layers with many rows (the sequence models' QKV / output / feed-forward projections) on the bf16x6 path:
# packed input, epilogue-fused bias / activation / dropout, packed g_z from the activation backward, bf16x6 input and weight
# gradients.  BL_LINEAR_X6=0: exact-fp32 MFMA GEMMs.
LINEAR_X6 = os.environ.get("BL_LINEAR_X6", "1") != "0"
LINEAR_X6_MIN_ROWS = 1024


def act_bwd_packed(g_y, y, act, drop: "Dropout", g_bias=None):
    """-> bf16x3-packed g_z int16 [R, 3 N] of y = drop(act(z + bias)) (bl_act_bwd_packed); g_bias accumulates column sums."""
    R, N = y.shape
    out = torch.empty((R, 3 * N), dtype=torch.int16, device=y.device)
    _check(load_library().bl_act_bwd_packed(_f32(g_y).data_ptr(), _f32(y).data_ptr(), R, N, y.stride(0), int(act), drop.c(), None, _p(g_bias),
                                            out.data_ptr(), _stream()), "bl_act_bwd_packed")
    return out


class _GatherLinear(torch.autograd.Function):
    """act(concat_j(X_j[idx_j]) @ W + b) without materialising the gather/concat."""

    @staticmethod
    def forward(ctx, W, bias, act, nsrc, *flat):
        drop = NO_DROPOUT
        if len(flat) == 2 * nsrc + 1:  # optional trailing Dropout: y = drop(act(x W + b))
            drop, flat = flat[-1], flat[:-1]
        xs, idxs = flat[:nsrc], flat[nsrc:]
        sources = list(zip(xs, idxs))
        R = idxs[0].shape[0] if idxs[0] is not None else xs[0].shape[0]
        K, N = W.shape
        x6 = (LINEAR_X6 and GEMM_MODE == "bf16x6" and nsrc == 1 and idxs[0] is None and R >= LINEAR_X6_MIN_ROWS and K % 32 == 0
              and N % 32 == 0 and act in (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH) and xs[0].is_contiguous())
        if x6:
            need_bwd = any(ctx.needs_input_grad)
            xp = pack_bf16x3(xs[0])
            wkn, wnk = _packed_layer_weights(_f32(W, "W"), need_bwd)
            out = gemm_rows_x6([(xp, None, K)], wkn, R, N, bias=bias, act=act, drop=drop, kind="linear_x6")
            # (the OUTPUT goes through save_for_backward: kept as a plain ctx attribute it forms the cycle output -> grad_fn -> ctx ->
            # output, which Python's collector cannot see through the C++ node -- every step's activations stayed allocated,
            # ~1 GiB per seq-great step until the device was full)
            ctx.save_for_backward(out)
            ctx.saved = (W, bias, act, sources, drop, xp if need_bwd else None, wnk)
            return out
        out = gemm_rows(sources, _f32(W, "W"), R, W.shape[1], bias=bias, act=act, drop=drop)
        ctx.save_for_backward(out)
        ctx.saved = (W, bias, act, sources, drop, None, None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        W, bias_p, act, sources, drop, xp, wnk = _take_saved(ctx)
        (out,) = ctx.saved_tensors
        has_bias = bias_p is not None
        R, N = out.shape
        K = W.shape[0]
        dev = W.device
        # weight / bias gradients are accumulated by the kernels: straight into .grad where the optimiser opted in (FlatAdam's flat
        # buffer: no zero fill, no autograd accumulation kernel -- 20 Linears per step in seq-great), into fresh zeros otherwise
        g_W, r_W = _grad_target(W)
        g_bias, r_bias = _grad_target(bias_p) if has_bias else (None, None)
        if xp is not None:  # bf16x6 path
            gzp = act_bwd_packed(g_out.contiguous(), out, act, drop, g_bias)
            gemm_wgrad_x6([(xp, None, K)], gzp, R, N, g_W)
            g_x = gemm_rows_x6([(gzp, None, N)], wnk, R, K, kind="linear_dgrad_x6") if ctx.needs_input_grad[4] else None
            return (r_W, r_bias, None, None, g_x, None) + ((None,) if drop is not NO_DROPOUT else ())
        g_z = act_bwd(g_out.contiguous(), out, act, drop, g_bias)
        gemm_wgrad(sources, g_z, R, N, g_W)
        g_a = gemm_rows([(g_z, None)], W, R, K, b_is_nk=True, ldb=N)
        g_xs, off = [], 0
        for j, (x, idx) in enumerate(sources):
            w = x.shape[1]
            if not ctx.needs_input_grad[4 + j]:
                g_xs.append(None)
            elif idx is None:
                g_xs.append(g_a[:, off : off + w].contiguous())
            else:
                g_x = torch.zeros_like(x)
                scatter_add_rows(g_a, off, w, idx, g_x)
                g_xs.append(g_x)
            off += w
        return (r_W, r_bias, None, None) + tuple(g_xs) + (None,) * (len(sources) + (1 if drop is not NO_DROPOUT else 0))


def gather_linear(sources: Sequence[RowSource], W, bias, act: str = "none", drop: Dropout = NO_DROPOUT):
    """drop(act(concat_j(x_j[idx_j]) @ W + bias)).  W / bias gradients: added straight into `W.grad` / `bias.grad` (backward
    returns None for them) when the parameter opted in -- see _opted_in_for_direct_grad for the contract and its consequences
    for torch.autograd.grad and parameter hooks -- through autograd otherwise."""
    xs = [x for x, _ in sources]
    idxs = [i for _, i in sources]
    extra = (drop,) if drop is not NO_DROPOUT else ()
    return _GatherLinear.apply(W, bias, _ACTS[act], len(sources), *xs, *idxs, *extra)


class _RowDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        R, H = x.shape
        y = torch.empty((R,), dtype=torch.float32, device=x.device)
        _check(load_library().bl_rowdot_fwd(_f32(x).data_ptr(), x.stride(0), _f32(w).data_ptr(), _p(b), R, H, y.data_ptr(), _stream()),
               "bl_rowdot_fwd")
        ctx.saved = (x, w, b is not None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, w, has_b = _take_saved(ctx)
        R, H = x.shape
        g_x = torch.empty_like(x)
        g_w = torch.zeros_like(w)
        g_b = torch.zeros((1,), dtype=torch.float32, device=x.device) if has_b else None
        _check(
            load_library().bl_rowdot_bwd(_f32(g_y.contiguous()).data_ptr(), x.data_ptr(), x.stride(0), w.data_ptr(), R, H,
                                         g_x.data_ptr(), g_x.stride(0), g_w.data_ptr(), _p(g_b), _stream()),
            "bl_rowdot_bwd")
        return g_x, g_w, g_b


def rowdot(x, w, b=None):
    return _RowDot.apply(x, w, b)


class _SegmentLogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr, seg_items, nseg, eps):
        y = torch.empty_like(x)
        _check(
            load_library().bl_segment_log_softmax_fwd(_f32(x).data_ptr(), _i32(seg_ptr).data_ptr(), _p(seg_items), int(nseg),
                                                      float(eps), y.data_ptr(), _stream()),
            "bl_segment_log_softmax_fwd")
        ctx.save_for_backward(y)  # (an output: see _GatherLinear)
        ctx.saved = (seg_ptr, seg_items, nseg)
        return y

    @staticmethod
    def backward(ctx, g_y):
        seg_ptr, seg_items, nseg = _take_saved(ctx)
        (y,) = ctx.saved_tensors
        g_x = torch.zeros_like(y)
        _check(
            load_library().bl_segment_log_softmax_bwd(_f32(g_y.contiguous()).data_ptr(), y.data_ptr(), seg_ptr.data_ptr(),
                                                      _p(seg_items), int(nseg), g_x.data_ptr(), _stream()),
            "bl_segment_log_softmax_bwd")
        return g_x, None, None, None, None


def segment_log_softmax(x, seg_ptr, seg_items, nseg: int, eps: float = 1e-12):
    """scatter_log_softmax (reference buglab/models/utils.py:15-28) over a CSR of the segment ids."""
    if x.numel() == 0:
        return x
    return _SegmentLogSoftmax.apply(x.contiguous(), seg_ptr, seg_items, nseg, eps)


class _SegmentMaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr, seg_of, nseg):
        out, arg, _, _, _ = segment_max(x, seg_ptr, None, nseg)
        ctx.saved = (arg, x.shape, seg_of)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g_out, _g_arg):
        arg, shape, seg_of = _take_saved(ctx)
        x_like = torch.empty(shape, dtype=torch.float32, device=g_out.device)
        g_x = segment_max_bwd(g_out.contiguous(), arg, x_like, seg_of, out=x_like)
        return g_x, None, None, None


def segment_max_pool(x, seg_ptr, seg_of, nseg: int):
    """scatter_max over CONTIGUOUS segments (rows of segment s are seg_ptr[s]..seg_ptr[s+1]).
    Returns (values [nseg, D], argmax row int32 [nseg, D]; -1 for an empty segment)."""
    return _SegmentMaxPool.apply(x.contiguous(), seg_ptr, seg_of, nseg)



# ------------------------------------------------------------------------------------------------
# whole scoring heads per C call (csrc/bl_heads_fused.hip)
def _grad_target(param):
    """(buffer the kernels accumulate into, what backward returns for it)."""
    d = _direct_small(param)
    if d is not None:
        return d, None
    z = torch.zeros_like(param)
    return z, z


class _MlpScore(torch.autograd.Function):
    """score[r] = w2 . relu(concat_j(x_j[idx_j[r]]) @ W1 + b1) + b2: forward and backward are one C call each."""

    @staticmethod
    def forward(ctx, W1, b1, w2, b2, nsrc, *flat):
        xs, idxs = flat[:nsrc], flat[nsrc:]
        rows, K = _rows(list(zip(xs, idxs)))
        R = idxs[0].shape[0] if idxs[0] is not None else xs[0].shape[0]
        H = W1.shape[1]
        dev = W1.device
        hidden = torch.empty((R, H), dtype=torch.float32, device=dev)
        score = torch.empty((R,), dtype=torch.float32, device=dev)
        _check(load_library().bl_gather_concat_mlp_score_fwd(ctypes.byref(rows), _f32(W1, "W1").data_ptr(), _f32(b1).data_ptr(),
                                                             _f32(w2).data_ptr(), _p(b2), R, H, hidden.data_ptr(), score.data_ptr(),
                                                             _stream()), "bl_gather_concat_mlp_score_fwd")
        ctx.saved = (W1, b1, w2, b2, xs, idxs, hidden, K)
        return score

    @staticmethod
    def backward(ctx, g_score):
        W1, b1, w2, b2, xs, idxs, hidden, K = _take_saved(ctx)
        lib = load_library()
        R, H = hidden.shape
        dev = W1.device
        rows, _ = _rows(list(zip(xs, idxs)))
        (gW1, rW1), (gb1, rb1), (gw2, rw2) = _grad_target(W1), _grad_target(b1), _grad_target(w2)
        gb2, rb2 = _grad_target(b2) if b2 is not None else (None, None)
        # one gradient matrix per DISTINCT source tensor (the scorers read the same node-state matrix two or three times)
        bufs, ret = {}, []
        gx = (c_void_p * 3)()
        ld = (c_int32 * 3)()
        for j, x in enumerate(xs):
            if not ctx.needs_input_grad[5 + j]:
                ret.append(None)
                continue
            key = x.data_ptr()
            if key not in bufs:
                bufs[key] = torch.zeros_like(x)
                ret.append(bufs[key])
            else:
                ret.append(None)
            gx[j], ld[j] = bufs[key].data_ptr(), bufs[key].stride(0)
        ws = torch.empty((lib.bl_gather_concat_mlp_score_workspace_bytes(R, H, K),), dtype=torch.uint8, device=dev)
        _check(lib.bl_gather_concat_mlp_score_bwd(ctypes.byref(rows), W1.data_ptr(), w2.data_ptr(), hidden.data_ptr(),
                                                  _f32(g_score.contiguous()).data_ptr(), R, H, ws.data_ptr(), gW1.data_ptr(), gb1.data_ptr(),
                                                  gw2.data_ptr(), _p(gb2), gx, ld, _stream()), "bl_gather_concat_mlp_score_bwd")
        return (rW1, rb1, rw2, rb2, None) + tuple(ret) + (None,) * len(xs)


def mlp_score(sources: Sequence[RowSource], W1, b1, w2, b2):
    xs = [x for x, _ in sources]
    idxs = [i for _, i in sources]
    return _MlpScore.apply(W1, b1, w2, b2, len(sources), *xs, *idxs)


class _LocalizationScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cand, cand_graph, cand_ptr, B, Ws, bs, W1, b1, w):
        lib = load_library()
        _f32(x, "node states")
        C, H = cand.shape[0], x.shape[1]
        dev = x.device
        saved = torch.empty((lib.bl_localization_scores_saved_bytes(C, B, H),), dtype=torch.uint8, device=dev)
        ws = torch.empty((lib.bl_localization_scores_workspace_bytes(C, B, H, 0),), dtype=torch.uint8, device=dev)
        score = torch.empty((C,), dtype=torch.float32, device=dev)
        _check(lib.bl_localization_scores_fwd(x.data_ptr(), x.stride(0), _i32(cand).data_ptr(), _i32(cand_graph).data_ptr(),
                                              _i32(cand_ptr).data_ptr(), C, B, H, _f32(Ws).data_ptr(), _f32(bs).data_ptr(),
                                              _f32(W1).data_ptr(), _f32(b1).data_ptr(), _f32(w).data_ptr(), saved.data_ptr(),
                                              ws.data_ptr(), score.data_ptr(), _stream()), "bl_localization_scores_fwd")
        ctx.saved = (x, cand, cand_graph, cand_ptr, B, Ws, bs, W1, b1, w, saved)
        return score

    @staticmethod
    def backward(ctx, g_score):
        x, cand, cand_graph, cand_ptr, B, Ws, bs, W1, b1, w, saved = _take_saved(ctx)
        lib = load_library()
        C, H = cand.shape[0], x.shape[1]
        dev = x.device
        g_x = torch.zeros_like(x)
        (gWs, rWs), (gbs, rbs), (gW1, rW1), (gb1, rb1), (gw, rw) = (_grad_target(t) for t in (Ws, bs, W1, b1, w))
        ws = torch.empty((lib.bl_localization_scores_workspace_bytes(C, B, H, 1),), dtype=torch.uint8, device=dev)
        _check(lib.bl_localization_scores_bwd(x.data_ptr(), x.stride(0), cand.data_ptr(), cand_graph.data_ptr(), cand_ptr.data_ptr(), C, B, H,
                                              Ws.data_ptr(), W1.data_ptr(), w.data_ptr(), saved.data_ptr(), ws.data_ptr(),
                                              _f32(g_score.contiguous()).data_ptr(), g_x.data_ptr(), g_x.stride(0), gWs.data_ptr(),
                                              gbs.data_ptr(), gW1.data_ptr(), gb1.data_ptr(), gw.data_ptr(), _stream()),
               "bl_localization_scores_bwd")
        return g_x, None, None, None, None, rWs, rbs, rW1, rb1, rw


def localization_scores(x, cand, cand_graph, cand_ptr, num_graphs: int, Ws, bs, W1, b1, w):
    """Candidate scores of the localization head before the NO_BUG logit (reference localizationmodule.py:54-60)."""
    if cand.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.float32, device=x.device)
    return _LocalizationScores.apply(x.contiguous(), cand, cand_graph, cand_ptr, int(num_graphs), Ws, bs, W1, b1, w)


# ------------------------------------------------------------------------------------------------
# loss assembly (csrc/bl_loss.hip): everything between the scorers' logits and the scalar loss in one kernel per direction
FUSED_LOSS = os.environ.get("BL_FUSED_LOSS", "1") != "0"
BUG_LOSS_STATS = 16


class BugLossIndex(NamedTuple):
    """Index tensors of one minibatch the loss assembly reads (all int32 on the device, has_bug bool)."""

    loc_group_ptr: torch.Tensor
    loc_group_items: torch.Tensor
    candidate_ptr: torch.Tensor
    has_bug: torch.Tensor
    correct_candidate_idxs: torch.Tensor
    repair_group_ptr: torch.Tensor
    repair_group_items: torch.Tensor
    logit_groups: tuple   # (text, var, swap): location group of every logit
    targets: tuple        # (text, var, swap): indices of the correct rewrites inside their slice
    num_groups: int


def _bug_loss_desc(loc_scores, logits, sizes, ix: BugLossIndex, w_buggy: float, abstain: float) -> bl_bug_loss_t:
    d = bl_bug_loss_t()
    d.B, d.C = int(ix.has_bug.shape[0]), int(loc_scores.shape[0])
    d.Rt, d.Rv, d.Rs = (int(n) for n in sizes)
    d.G = int(ix.num_groups)
    d.loc_scores, d.repair_logits = _p(loc_scores), _p(logits)
    d.loc_group_ptr, d.loc_group_items = _i32(ix.loc_group_ptr).data_ptr(), _i32(ix.loc_group_items).data_ptr()
    d.candidate_ptr = _i32(ix.candidate_ptr).data_ptr()
    d.has_bug = _req(ix.has_bug, torch.bool, "has_bug").data_ptr()
    d.correct_candidate_idxs = _i32(ix.correct_candidate_idxs).data_ptr()
    d.repair_group_ptr, d.repair_group_items = _p(ix.repair_group_ptr), _p(ix.repair_group_items)
    for k in range(3):
        d.logit_group[k] = _i32(ix.logit_groups[k]).data_ptr() if ix.logit_groups[k].numel() else None
        d.target[k] = _i32(ix.targets[k]).data_ptr() if ix.targets[k].numel() else None
        d.ntarget[k] = int(ix.targets[k].shape[0])
    d.w_buggy, d.abstain_weight = float(w_buggy), float(abstain)
    return d


class _BugLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc_scores, logits, sizes, ix: BugLossIndex, w_buggy: float, abstain: float):
        _f32(loc_scores, "loc_scores")
        _f32(logits, "repair logits")
        dev = loc_scores.device
        d = _bug_loss_desc(loc_scores, logits, sizes, ix, w_buggy, abstain)
        loc_lp = torch.empty((d.C + d.B,), dtype=torch.float32, device=dev)
        rep_lp = torch.empty((max(1, logits.shape[0]),), dtype=torch.float32, device=dev)
        gmax = torch.empty((max(1, d.G),), dtype=torch.float32, device=dev)
        out = torch.empty((1 + BUG_LOSS_STATS,), dtype=torch.float32, device=dev)  # [loss | stats]
        _check(load_library().bl_bug_loss_fwd(ctypes.byref(d), loc_lp.data_ptr(), rep_lp.data_ptr(), gmax.data_ptr(), out.data_ptr(),
                                              out[1:].data_ptr(), _stream()), "bl_bug_loss_fwd")
        ctx.saved = (loc_scores, logits, sizes, ix, w_buggy, abstain, loc_lp, rep_lp)
        loss, stats = out[0], out[1:]
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        loc_scores, logits, sizes, ix, w_buggy, abstain, loc_lp, rep_lp = _take_saved(ctx)
        dev = loc_scores.device
        d = _bug_loss_desc(loc_scores, logits, sizes, ix, w_buggy, abstain)
        scratch = torch.empty((d.C + d.B + logits.shape[0] + 1,), dtype=torch.float32, device=dev)
        g_scores = torch.empty_like(loc_scores)
        g_logits = torch.empty_like(logits)
        g = g_loss.contiguous().reshape(1)
        _check(load_library().bl_bug_loss_bwd(ctypes.byref(d), loc_lp.data_ptr(), rep_lp.data_ptr(), _f32(g).data_ptr(), scratch.data_ptr(),
                                              _p(g_scores), _p(g_logits), _stream()), "bl_bug_loss_bwd")
        return g_scores, g_logits, None, None, None, None


def bug_loss(loc_scores, logits, sizes, ix: BugLossIndex, w_buggy: float = 1.0, abstain_weight: float = 0.0):
    """-> (loss scalar, stats [16]); see include/buglab_hip.h::bl_bug_loss_t.  logits = cat(text, var, swap) with `sizes` rows each."""
    return _BugLoss.apply(loc_scores.contiguous(), logits.contiguous(), tuple(int(n) for n in sizes), ix, float(w_buggy), float(abstain_weight))


# ------------------------------------------------------------------------------------------------
# `seq-great` relational transformer block (csrc/bl_seq_ops.hip + the MFMA GEMMs)
class _AddLayerNorm(torch.autograd.Function):
    """y = LayerNorm(x + r) (r optional); backward hands the same gradient to x and r."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps):
        _f32(x, "x")
        n, D = x.shape
        dev = x.device
        z = torch.empty_like(x) if r is not None else x
        y = torch.empty_like(x)
        mean = torch.empty((n,), dtype=torch.float32, device=dev)
        rstd = torch.empty((n,), dtype=torch.float32, device=dev)
        _check(load_library().bl_add_layernorm_fwd(x.data_ptr(), _p(r), _f32(gamma).data_ptr(), _f32(beta).data_ptr(), float(eps), n, D,
                                                   z.data_ptr() if r is not None else None, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   _stream()), "bl_add_layernorm_fwd")
        ctx.saved = (z, mean, rstd, gamma, beta, r is not None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        z, mean, rstd, gamma, beta, has_r = _take_saved(ctx)
        (gg, rg), (gb, rb) = _grad_target(gamma), _grad_target(beta)
        g_z = layernorm_bwd(g_y.contiguous(), z, mean, rstd, gamma, gg, gb)
        return g_z, (g_z if has_r else None), rg, rb, None


def add_layernorm(x, r, gamma, beta, eps: float = 1e-5):
    return _AddLayerNorm.apply(x.contiguous(), r.contiguous() if r is not None else None, gamma, beta, eps)


class RelEdges(NamedTuple):
    """Edges of a padded [B, L] minibatch as a CSR over query rows b * L + i (buglab.data.seqcollate.edge_csr)."""

    row_ptr: torch.Tensor   # int32 [B * L + 1]
    key: torch.Tensor       # int32 [n]  key position of the entry
    code: torch.Tensor      # int32 [n]  2 * edge_type + direction (0: the query is the edge's source, 1: its target)
    num_entries: int


_group_ptr_cache = {}


def _uniform_group_ptr(G: int, L: int, device):
    key = (G, L, str(device))
    t = _group_ptr_cache.get(key)
    if t is None:
        if len(_group_ptr_cache) > 64:
            _group_ptr_cache.clear()
        t = (torch.arange(G + 1, dtype=torch.int64) * L).to(torch.int32).to(device)
        _group_ptr_cache[key] = t
    return t


FUSED_ATTENTION = os.environ.get("BL_FUSED_ATTENTION", "1") != "0"  # seq-great: scores -> probabilities in one kernel


class _RelAttention(torch.autograd.Function):
    """Relational multi-head self-attention between the QKV projection and the output projection
    (reference multihead_attention.py:46-80, relational_multihead_attention.py:72-178).  Q.K^T, P.V and their four
    gradient products are MFMA GEMMs grouped by (sample, head); edge terms, masked softmax and value biases are the
    row-wise kernels of csrc/bl_seq_ops.hip."""

    @staticmethod
    def forward(ctx, qkv, lens, edges: RelEdges, bias_f, bias_r, vb_f, vb_r, B, L, H, dk, T, scalar_bias, drop: Dropout):
        lib = load_library()
        _f32(qkv, "qkv")
        G, D = B * H, H * dk
        st = _stream()
        scale = float(dk) ** -0.5
        t3 = qkv.view(B, L, H, 3, dk).permute(3, 0, 2, 1, 4).contiguous()  # [3, B, H, L, dk]
        qs, kt, vt = t3[0], t3[1], t3[2]
        qs.mul_(scale)  # multihead_attention.py:54: queries pre-scaled
        gptr = _uniform_group_ptr(G, L, qkv.device)
        mode = 1 if scalar_bias else 0
        has_e = edges.num_entries > 0
        if FUSED_ATTENTION and mode == 0 and lib.bl_rel_attn_probs_ok(L, dk, T):
            # scores, edge terms, masked softmax and nn.Dropout in one kernel: the scores never reach memory
            P = torch.empty((G * L, L), dtype=torch.float32, device=qkv.device)
            Pd = torch.empty_like(P) if drop.p > 0 else P
            with _timed("attn_probs_fwd", 0.0, nbytes=4.0 * G * L * (L * (2 if drop.p > 0 else 1) + 2 * dk)):  # writes P (+ Pd), reads q, k
                _check(lib.bl_rel_attn_probs_fwd(qs.data_ptr(), kt.data_ptr(), edges.row_ptr.data_ptr() if has_e else None,
                                                 edges.key.data_ptr() if has_e else None, edges.code.data_ptr() if has_e else None, B, L, H, dk, T,
                                                 _f32(bias_f).data_ptr(), _f32(bias_r).data_ptr(), _i32(lens).data_ptr(), drop.c(), P.data_ptr(),
                                                 Pd.data_ptr(), st), "bl_rel_attn_probs_fwd")
        else:
            S = gemm_rows([(qs.view(G * L, dk), None)], kt, G * L, L, b_is_nk=True, b_group_stride=L * dk, ldb=dk, group_ptr=gptr, G=G)
            if has_e:
                _check(lib.bl_rel_attn_bias_fwd(edges.row_ptr.data_ptr(), edges.key.data_ptr(), edges.code.data_ptr(), B, L, H, dk, mode,
                                                (kt if scalar_bias else qs).data_ptr(), _f32(bias_f).data_ptr(), _f32(bias_r).data_ptr(),
                                                S.data_ptr(), st), "bl_rel_attn_bias_fwd")
            P = S
            # softmax and nn.Dropout on the probabilities (multihead_attention.py:65-72) in one pass over the scores
            Pd = torch.empty_like(P) if drop.p > 0 else P
            _check(lib.bl_masked_softmax_dropout_fwd(S.data_ptr(), G * L, L, H * L, _i32(lens).data_ptr(), drop.c(), Pd.data_ptr(), st),
                   "bl_masked_softmax_dropout_fwd")
        mm32 = bool(FUSED_ATTENTION and lib.bl_attn_mm32_ok(L, dk))  # the skinny products on their own kernels (head dimension 32)
        if mm32:
            ctx_t = torch.empty((G * L, dk), dtype=torch.float32, device=qkv.device)
            with _timed("attn_rows_times", 2.0 * G * L * L * dk, nbytes=4.0 * G * L * (L + 2 * dk)):
                _check(lib.bl_attn_rows_times(Pd.data_ptr(), vt.data_ptr(), G, L, dk, None, 1.0, ctx_t.data_ptr(), st), "bl_attn_rows_times")
        else:
            ctx_t = gemm_rows([(Pd, None)], vt, G * L, dk, b_group_stride=L * dk, ldb=dk, group_ptr=gptr, G=G)
        if vb_f is not None and edges.num_entries > 0:
            _check(lib.bl_rel_value_bias_fwd(edges.row_ptr.data_ptr(), edges.key.data_ptr(), edges.code.data_ptr(), B, L, H, dk,
                                             Pd.data_ptr(), _f32(vb_f).data_ptr(), _f32(vb_r).data_ptr(), ctx_t.data_ptr(), st),
                   "bl_rel_value_bias_fwd")
        out = ctx_t.view(B, H, L, dk).permute(0, 2, 1, 3).contiguous().view(B * L, D)
        ctx.fused = bool(FUSED_ATTENTION and mode == 0 and vb_f is None and lib.bl_rel_attn_probs_ok(L, dk, T))
        ctx.mm32 = mm32
        ctx.saved = (qs, kt, vt, P, Pd, lens, edges, bias_f, bias_r, vb_f, vb_r, B, L, H, dk, T, mode, drop, gptr, scale)
        return out

    @staticmethod
    def backward(ctx, g_out):
        qs, kt, vt, P, Pd, lens, edges, bias_f, bias_r, vb_f, vb_r, B, L, H, dk, T, mode, drop, gptr, scale = _take_saved(ctx)
        lib = load_library()
        G, D = B * H, H * dk
        dev = g_out.device
        st = _stream()
        g_ct = g_out.view(B, L, H, dk).permute(0, 2, 1, 3).contiguous().view(G * L, dk)
        mm32 = ctx.mm32
        g3 = (torch.empty if mm32 else torch.zeros)((3, B, H, L, dk), dtype=torch.float32, device=dev)
        g_qs, g_k, g_v = g3[0], g3[1], g3[2]

        def tn(a, bm, out):  # out[g] = a[g]^T . bm[g]
            if mm32:
                with _timed("attn_transposed_times", 2.0 * G * L * L * dk, nbytes=4.0 * G * L * (L + 2 * dk)):
                    _check(lib.bl_attn_transposed_times(a.data_ptr(), bm.data_ptr(), G, L, dk, out.data_ptr(), st), "bl_attn_transposed_times")
            else:
                gemm_wgrad([(a, None)], bm.view(G * L, dk), G * L, dk, out.view(G, L, dk), gw_group_stride=L * dk, group_ptr=gptr, G=G)

        tn(Pd, g_ct, g_v)
        has_e = edges.num_entries > 0
        ep = (edges.row_ptr.data_ptr(), edges.key.data_ptr(), edges.code.data_ptr()) if has_e else None
        if ctx.fused:
            # dO.V^T, dropout mask, softmax backward and the edge terms' gradients in one kernel; dS is written once
            (g_bf, r_bf), (g_br, r_br) = _grad_target(bias_f), _grad_target(bias_r)
            dS = torch.empty((G * L, L), dtype=torch.float32, device=dev)
            gq_edge = torch.empty((G * L, dk), dtype=torch.float32, device=dev) if has_e else None  # (the kernel writes every row)
            with _timed("attn_probs_bwd", 0.0, nbytes=4.0 * G * L * (2 * L + 3 * dk)):  # reads P, dO, v, q; writes dS
                _check(lib.bl_rel_attn_probs_bwd(g_ct.data_ptr(), vt.data_ptr(), P.data_ptr(), qs.data_ptr(), *(ep or (None, None, None)), B, L, H, dk, T,
                                                 bias_f.data_ptr(), bias_r.data_ptr(), drop.c(), dS.data_ptr(), _p(gq_edge), g_bf.data_ptr(),
                                                 g_br.data_ptr(), st), "bl_rel_attn_probs_bwd")
            if mm32:  # dQ = (dS.K + edge part) * scale in one kernel
                with _timed("attn_rows_times", 2.0 * G * L * L * dk, nbytes=4.0 * G * L * (L + 2 * dk)):
                    _check(lib.bl_attn_rows_times(dS.data_ptr(), kt.data_ptr(), G, L, dk, _p(gq_edge), scale, g_qs.data_ptr(), st), "bl_attn_rows_times")
            else:
                gemm_rows([(dS, None)], kt, G * L, dk, b_group_stride=L * dk, ldb=dk, group_ptr=gptr, G=G, out=g_qs.view(G * L, dk))
                if has_e:
                    g_qs.view(G * L, dk).add_(gq_edge)
                g_qs.mul_(scale)
            tn(dS, qs, g_k)
            g_qkv = g3.permute(1, 3, 2, 0, 4).contiguous().view(B * L, 3 * D)
            return g_qkv, None, None, r_bf, r_br, None, None, None, None, None, None, None, None, None
        dP = gemm_rows([(g_ct, None)], vt, G * L, L, b_is_nk=True, b_group_stride=L * dk, ldb=dk, group_ptr=gptr, G=G)
        r_vbf = r_vbr = None
        if vb_f is not None:
            (g_vbf, r_vbf), (g_vbr, r_vbr) = _grad_target(vb_f), _grad_target(vb_r)
            if has_e:
                _check(lib.bl_rel_value_bias_bwd(*ep, B, L, H, dk, T, Pd.data_ptr(), g_ct.data_ptr(), vb_f.data_ptr(), vb_r.data_ptr(),
                                                 dP.data_ptr(), g_vbf.data_ptr(), g_vbr.data_ptr(), st), "bl_rel_value_bias_bwd")
        _check(lib.bl_softmax_dropout_bwd(P.data_ptr(), dP.data_ptr(), G * L, L, drop.c(), st), "bl_softmax_dropout_bwd")  # (mask, then softmax')
        dS = dP
        if mm32:
            _check(lib.bl_attn_rows_times(dS.data_ptr(), kt.data_ptr(), G, L, dk, None, 1.0, g_qs.data_ptr(), st), "bl_attn_rows_times")
        else:
            gemm_rows([(dS, None)], kt, G * L, dk, b_group_stride=L * dk, ldb=dk, group_ptr=gptr, G=G, out=g_qs.view(G * L, dk))
        tn(dS, qs, g_k)
        (g_bf, r_bf), (g_br, r_br) = _grad_target(bias_f), _grad_target(bias_r)
        if has_e:
            _check(lib.bl_rel_attn_bias_bwd(*ep, B, L, H, dk, mode, T, (kt if mode == 1 else qs).data_ptr(), bias_f.data_ptr(),
                                            bias_r.data_ptr(), dS.data_ptr(), g_qs.data_ptr(), g_k.data_ptr(), g_bf.data_ptr(),
                                            g_br.data_ptr(), st), "bl_rel_attn_bias_bwd")
        g_qs.mul_(scale)
        g_qkv = g3.permute(1, 3, 2, 0, 4).contiguous().view(B * L, 3 * D)
        return g_qkv, None, None, r_bf, r_br, r_vbf, r_vbr, None, None, None, None, None, None, None


def rel_attention(qkv, lens, edges: RelEdges, bias_f, bias_r, vb_f, vb_r, B, L, H, dk, T, scalar_bias=False, drop: Dropout = NO_DROPOUT):
    """qkv [B*L, H*3*dk] (per head [q | k | v]) -> attention context [B*L, H*dk]."""
    return _RelAttention.apply(qkv.contiguous(), lens, edges, bias_f, bias_r, vb_f, vb_r, int(B), int(L), int(H), int(dk), int(T),
                               bool(scalar_bias), drop)


# ---- `seq-gru`: the time recurrence of one bidirectional GRU layer (csrc/bl_gru_scan.hip) -------------------------------------
class _GruScan(torch.autograd.Function):
    """gi [B L, 6 Hh] (x W_ih + b_ih of both directions, columns [direction][r | z | n]) -> h_t of both directions [B L, 2 Hh] with
    torch.nn.GRU's PackedSequence semantics (reference seqmodel.py:385-392).  Backward: one reverse scan (bl_gru_scan_bwd) gives the
    gradient of gi and of the recurrent pre-activations; the recurrent weight gradient h_prev^T d_gh is a weight-gradient GEMM."""

    @staticmethod
    def forward(ctx, gi, W_hh, b_hh, lens, B, L):
        lib = load_library()
        _f32(gi, "gi")
        Hh = W_hh.shape[1]
        R = B * L
        assert gi.shape == (R, 6 * Hh) and W_hh.shape == (2, Hh, 3 * Hh) and b_hh.shape == (2, 3 * Hh)
        need_bwd = any(ctx.needs_input_grad)
        out = torch.empty((R, 2 * Hh), dtype=torch.float32, device=gi.device)
        saved = torch.empty((lib.bl_gru_scan_saved_elems(B, L, Hh),), dtype=torch.float32, device=gi.device) if need_bwd else None
        _check(lib.bl_gru_scan_fwd(gi.data_ptr(), gi.stride(0), _f32(W_hh.contiguous()).data_ptr(), _f32(b_hh.contiguous()).data_ptr(),
                                   _i32(lens).data_ptr(), B, L, Hh, out.data_ptr(), out.stride(0), _p(saved), _stream()), "bl_gru_scan_fwd")
        ctx.saved = (W_hh, lens, B, L, Hh, saved)
        return out

    @staticmethod
    def backward(ctx, g_out):
        W_hh, lens, B, L, Hh, saved = _take_saved(ctx)
        lib = load_library()
        R = B * L
        dev = g_out.device
        g_out = g_out.contiguous()
        g_gi = torch.empty((R, 6 * Hh), dtype=torch.float32, device=dev)
        g_gh = torch.empty((2, R, 3 * Hh), dtype=torch.float32, device=dev)
        _check(lib.bl_gru_scan_bwd(g_out.data_ptr(), g_out.stride(0), _f32(W_hh.contiguous()).data_ptr(), saved.data_ptr(), _i32(lens).data_ptr(),
                                   B, L, Hh, g_gi.data_ptr(), g_gi.stride(0), g_gh.data_ptr(), _stream()), "bl_gru_scan_bwd")
        g_W = torch.zeros_like(W_hh)
        h_prev = saved[2 * R * 4 * Hh:].view(2, R, Hh)
        for d in range(2):
            gemm_wgrad([(h_prev[d], None)], g_gh[d], R, 3 * Hh, g_W[d])  # h_prev^T . d_gh
        return g_gi, g_W, g_gh.sum(1), None, None, None


def gru_scan(gi, W_hh, b_hh, lens, B: int, L: int):
    return _GruScan.apply(gi.contiguous(), W_hh, b_hh, lens, int(B), int(L))


# ---- one relational transformer encoder layer per C call (csrc/bl_great_layer.hip) ----------------------------------------
FUSED_GREAT_LAYER = os.environ.get("BL_FUSED_GREAT_LAYER", "1") != "0"  # A/B switch: 0 = the op-by-op path above


def great_layer_ok(B: int, L: int, H: int, dk: int, T: int, FF: int) -> bool:
    """Whether bl_great_layer_fwd / _bwd take the shape (the caller also checks the layer's configuration: postnorm, rezero
    off, vector query bias, no value biases)."""
    return bool(FUSED_GREAT_LAYER and LINEAR_X6 and GEMM_MODE == "bf16x6"
                and load_library().bl_great_layer_ok(int(B), int(L), int(H), int(dk), int(T), int(FF)))


def _great_desc(B, L, H, dk, T, FF, lens, edges: "RelEdges", bias_f, bias_r, norm_g, norm_b, lin1_b, lin2_b, packs, drops) -> bl_great_layer_t:
    d = bl_great_layer_t()
    d.B, d.L, d.H, d.dk, d.T, d.FF = int(B), int(L), int(H), int(dk), int(T), int(FF)
    if edges.num_entries > 0:
        d.row_ptr, d.ekey, d.ecode = _i32(edges.row_ptr).data_ptr(), _i32(edges.key).data_ptr(), _i32(edges.code).data_ptr()
    d.lens = _i32(lens).data_ptr()
    d.bias_f, d.bias_r = _f32(bias_f).data_ptr(), _f32(bias_r).data_ptr()
    d.norm_g, d.norm_b, d.lin1_b, d.lin2_b = _f32(norm_g).data_ptr(), _f32(norm_b).data_ptr(), _f32(lin1_b).data_ptr(), _f32(lin2_b).data_ptr()
    (qkv, qkv_b), (out, out_b), (l1, l1_b), (l2, l2_b) = packs
    d.qkv_w, d.out_w, d.lin1_w, d.lin2_w = qkv.data_ptr(), out.data_ptr(), l1.data_ptr(), l2.data_ptr()
    d.qkv_w_bwd, d.out_w_bwd, d.lin1_w_bwd, d.lin2_w_bwd = _p(qkv_b), _p(out_b), _p(l1_b), _p(l2_b)
    d.ln_eps = 1e-5
    d.drop_attn, d.drop_att_out, d.drop_ff_hidden, d.drop_ff_out = (x.c() for x in drops)
    return d


class _GreatLayer(torch.autograd.Function):
    """RelationalTransformerEncoderLayer.forward ("postnorm", rezero off, vector query bias) = one C call forward, one backward.
    `chain` carries the packed form of the activations from layer to layer: chain["packed"] is bl_pack_bf16x3 of THIS layer's
    input if chain["of"] is that tensor's address (written by the previous layer's call), and is replaced by the packed output."""

    @staticmethod
    def forward(ctx, x, qkv_W, out_W, bias_f, bias_r, lin1_W, lin1_b, lin2_W, lin2_b, norm_g, norm_b, lens, edges, dims, drops, chain):
        lib = load_library()
        _f32(x, "x")
        B, L, H, dk, T, FF = dims
        R, D = x.shape
        dev = x.device
        need_bwd = any(ctx.needs_input_grad)
        packs = [_packed_layer_weights(_f32(W, "W"), need_bwd) for W in (qkv_W, out_W, lin1_W, lin2_W)]
        d = _great_desc(B, L, H, dk, T, FF, lens, edges, bias_f, bias_r, norm_g, norm_b, lin1_b, lin2_b, packs, drops)
        xp = chain.get("packed") if (chain is not None and chain.get("of") == (x.data_ptr(), x._version)) else None
        saved = (torch.empty((lib.bl_great_layer_saved_bytes(B, L, H, dk, FF, 1 if xp is None else 0),),
                             dtype=torch.uint8, device=dev) if need_bwd else None)
        ws = torch.empty((lib.bl_great_layer_workspace_bytes(B, L, H, dk, FF, 0 if need_bwd else 3),), dtype=torch.uint8, device=dev)
        out = torch.empty_like(x)
        outp = torch.empty((R, 3 * D), dtype=torch.int16, device=dev) if chain is not None else None
        _check(lib.bl_great_layer_fwd(ctypes.byref(d), x.data_ptr(), _p(xp), out.data_ptr(), _p(outp), _p(saved), ws.data_ptr(), _stream()),
               "bl_great_layer_fwd")
        if chain is not None:
            chain["packed"], chain["of"] = outp, (out.data_ptr(), out._version)
        ctx.saved = (qkv_W, out_W, bias_f, bias_r, lin1_W, lin1_b, lin2_W, lin2_b, norm_g, norm_b, lens, edges, dims, drops, xp, saved, packs)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (qkv_W, out_W, bias_f, bias_r, lin1_W, lin1_b, lin2_W, lin2_b, norm_g, norm_b, lens, edges, dims, drops, xp, saved,
         packs) = _take_saved(ctx)
        lib = load_library()
        B, L, H, dk, T, FF = dims
        dev = g_out.device
        packs = [p if p[1] is not None else _packed_layer_weights(W, True) for p, W in zip(packs, (qkv_W, out_W, lin1_W, lin2_W))]
        d = _great_desc(B, L, H, dk, T, FF, lens, edges, bias_f, bias_r, norm_g, norm_b, lin1_b, lin2_b, packs, drops)
        params = (qkv_W, out_W, lin1_W, lin1_b, lin2_W, lin2_b, norm_g, norm_b, bias_f, bias_r)
        targets = [_grad_target(p) for p in params]
        g = bl_great_layer_grads_t()
        (g.qkv_w, g.out_w, g.lin1_w, g.lin1_b, g.lin2_w, g.lin2_b, g.norm_g, g.norm_b, g.bias_f, g.bias_r) = (t[0].data_ptr() for t in targets)
        ws = torch.empty((lib.bl_great_layer_workspace_bytes(B, L, H, dk, FF, 1),), dtype=torch.uint8, device=dev)
        g_x = torch.empty((B * L, H * dk), dtype=torch.float32, device=dev)
        side = _streams.side_stream_for_current_device()
        _check(lib.bl_great_layer_bwd(ctypes.byref(d), _p(xp), _f32(g_out.contiguous()).data_ptr(), saved.data_ptr(), ws.data_ptr(), g_x.data_ptr(),
                                      ctypes.byref(g), _stream(), side.cuda_stream if side is not None else None), "bl_great_layer_bwd")
        r = {id(p): t[1] for p, t in zip(params, targets)}
        return (g_x, r[id(qkv_W)], r[id(out_W)], r[id(bias_f)], r[id(bias_r)], r[id(lin1_W)], r[id(lin1_b)], r[id(lin2_W)], r[id(lin2_b)],
                r[id(norm_g)], r[id(norm_b)], None, None, None, None, None)


def great_layer(x, qkv_W, out_W, bias_f, bias_r, lin1_W, lin1_b, lin2_W, lin2_b, norm_g, norm_b, lens, edges: RelEdges, B, L, H, dk, T,
                drops=(NO_DROPOUT,) * 4, chain: Optional[dict] = None):
    """out = norm1(x1 + drop(linear2(drop(relu(linear1(x1)))))), x1 = norm1(x + drop(out_proj(rel_attention(qkv_proj(x))))) --
    reference relational_transformer.py:104-124 (postnorm; both sublayers normalised by norm1).  drops = (attention
    probabilities, attention branch, inside the feed-forward block, feed-forward branch)."""
    FF = lin1_W.shape[1]
    return _GreatLayer.apply(x.contiguous(), qkv_W, out_W, bias_f, bias_r, lin1_W, lin1_b, lin2_W, lin2_b, norm_g, norm_b, lens, edges,
                             (int(B), int(L), int(H), int(dk), int(T), int(FF)), tuple(drops), chain)


def dropout_rows(x, drop: Dropout):
    """Elementwise counter-hash dropout with autograd (embedding dropout of the sequence models)."""
    if drop.p <= 0:
        return x
    return _DropoutFn.apply(x, drop)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop):
        y = x.contiguous().clone()
        _check(load_library().bl_dropout_inplace(_f32(y).data_ptr(), y.numel(), drop.c(), _stream()), "bl_dropout_inplace")
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        _check(load_library().bl_dropout_inplace(g.data_ptr(), g.numel(), ctx.drop.c(), _stream()), "bl_dropout_inplace")
        return g, None

# ------------------------------------------------------------------------------------------------
# optimiser on flat buffers
_SQNORM_SCRATCH: dict = {}


def sqnorm(flat_grad: torch.Tensor, out: torch.Tensor, scratch: Optional[torch.Tensor] = None):
    """sum(g^2) -> out[0], added in one fixed order (replicas with equal gradients clip by the same number).  `scratch`
    (bl_sqnorm_scratch_bytes()) defaults to one buffer per (device, stream)."""
    lib = load_library()
    if scratch is None:
        key = (flat_grad.device, _stream())
        scratch = _SQNORM_SCRATCH.get(key)
        if scratch is None:
            scratch = _SQNORM_SCRATCH[key] = torch.empty(lib.bl_sqnorm_scratch_bytes() // 4, dtype=torch.float32, device=flat_grad.device)
    _check(lib.bl_sqnorm(_f32(flat_grad).data_ptr(), flat_grad.numel(), out.data_ptr(), scratch.data_ptr(), _stream()), "bl_sqnorm")
    return out


def adam_clip_step(param, grad, m, v, sqn, *, prescale=1.0, clip=0.5, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
    _check(
        load_library().bl_adam_clip_step(_f32(param).data_ptr(), _f32(grad).data_ptr(), _f32(m).data_ptr(), _f32(v).data_ptr(),
                                         param.numel(), _p(sqn), float(prescale), float(clip), float(lr), float(beta1), float(beta2),
                                         float(eps), int(step), _stream()),
        "bl_adam_clip_step")


def routed_dgrad_vec(gq, msg_tgt, win_bits, type_ptr, T, wt, E, K2):
    """g_a [E, K2] = routed message gradient x W^T from its non-zeros only (vector units; csrc/bl_routed_dgrad.hip).
    gq [N, Dm] fp32, wt [T, Dm, K2] = W transposed, win_bits [E, Dm/32] from segment_max."""
    Dm = gq.shape[1]
    out = torch.empty((E, K2), dtype=torch.float32, device=gq.device)
    with _timed("msg_dgrad_vec", 2.0 * gq.shape[0] * Dm * K2):
        _check(load_library().bl_routed_dgrad_vec(_f32(gq).data_ptr(), gq.stride(0), _i32(msg_tgt).data_ptr(), win_bits.data_ptr(),
                                                 win_bits.stride(0), _i32(type_ptr).data_ptr(), int(T), _f32(wt).data_ptr(), int(E), Dm,
                                                 int(K2), out.data_ptr(), out.stride(0), _stream()), "bl_routed_dgrad_vec")
    return out


def routed_dgrad_nodes(gq, msg_src, msg_tgt, win_bits, type_ptr, T, wt, E, Din, out_lo, out_hi=None, src_rows=None):
    """Adds the routed input gradient straight into the node gradient out_lo [N, split] (+ out_hi [N, Din - split]):
    routed_dgrad_vec + mp_scatter_grad without the per-message rows (fp32 atomics; the outputs must be zeroed).
    src_rows [E, Din]: the source half is written there per message instead (sum it with mp_scatter_grad, accumulate=1)."""
    Dm = gq.shape[1]
    split = out_lo.shape[1]
    with _timed("msg_dgrad_nodes", 2.0 * gq.shape[0] * Dm * 2 * Din):
        _check(load_library().bl_routed_dgrad_nodes_rows(_f32(gq).data_ptr(), gq.stride(0), _i32(msg_src).data_ptr(), _i32(msg_tgt).data_ptr(),
                                                        win_bits.data_ptr(), win_bits.stride(0), _i32(type_ptr).data_ptr(), int(T),
                                                        _f32(wt).data_ptr(), int(E), Dm, int(Din), int(split), out_lo.data_ptr(), out_lo.stride(0),
                                                        _p(out_hi), out_hi.stride(0) if out_hi is not None else 0, _p(src_rows),
                                                        src_rows.stride(0) if src_rows is not None else 0, _stream()),
               "bl_routed_dgrad_nodes_rows")
    return out_lo, out_hi


def adam_clip_step_dp(param, grad, m, v, sqn, batch_total, *, clip=0.5, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
    """Data-parallel form: grad = sum over ranks of B_rank * grad_rank, batch_total = device scalar sum of B_rank."""
    _check(
        load_library().bl_adam_clip_step_dp(_f32(param).data_ptr(), _f32(grad).data_ptr(), _f32(m).data_ptr(), _f32(v).data_ptr(),
                                            param.numel(), _p(sqn), _f32(batch_total).data_ptr(), float(clip), float(lr), float(beta1),
                                            float(beta2), float(eps), int(step), _stream()),
        "bl_adam_clip_step_dp")


# ------------------------------------------------------------------------------------------------
# Names whose one copy lives in a sub-module but that callers read / set on the package (bench.py, tests, tools):
#   hip_ops.USE_SIDE_STREAM / DIRECT_PARAM_GRAD / SIDE_STREAM_PRIORITY -> _streams;  hip_ops.CALL_COUNT, hip_ops._lib (the CDLL
#   handle: tools point it at a tuning build), hip_ops.LIB_PATH -> _lib
_FORWARDED = {"USE_SIDE_STREAM": _streams, "DIRECT_PARAM_GRAD": _streams, "SIDE_STREAM_PRIORITY": _streams,
              "CALL_COUNT": _lib_module, "_lib": _lib_module, "LIB_PATH": _lib_module}
for _name in _FORWARDED:
    globals().pop(_name, None)  # (`from ._lib import *` copied the values: the package must not hold stale ones)


class _HipOpsModule(types.ModuleType):
    def __getattr__(self, name):
        owner = _FORWARDED.get(name)
        if owner is None:
            raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")
        return getattr(owner, name)

    def __setattr__(self, name, value):
        owner = _FORWARDED.get(name)
        if owner is not None:
            setattr(owner, name, value)
        else:
            super().__setattr__(name, value)


sys.modules[__name__].__class__ = _HipOpsModule

#!/usr/bin/env python
"""Routed input gradient + node-gradient sums of one message-passing layer at the BASELINE c2 layer shape:
   (a) shipped: routed bf16x6 GEMM -> g_a [E, 2 Din] -> bl_mp_scatter_grad
   (b) non-zeros only on the vector units -> g_a -> bl_mp_scatter_grad
   (c) non-zeros only, node sums fused in (fp32 atomics, no g_a)
Checks (b), (c) against (a), then times each.  `--split` also exercises the two-output (ConcatResidual) form."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np
import torch

from buglab.models import hip_ops as ops


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def csr(idx, N):
    order = np.argsort(idx, kind="stable").astype(np.int32)
    ptr = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=N))]).astype(np.int32)
    return torch.from_numpy(ptr).cuda(), torch.from_numpy(order).cuda()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--din", type=int, default=128)
    ap.add_argument("--dm", type=int, default=128)
    ap.add_argument("--types", type=int, default=16)
    ap.add_argument("--degree", default="uniform")
    a = ap.parse_args()
    n_per, e_per = 2000, 10000
    N, E, Din, Dm, T = a.graphs * n_per, a.graphs * e_per, a.din, a.dm, a.types
    rng = np.random.default_rng(0)
    w = 1.0 / np.arange(1, T + 1)
    sizes = np.floor(w / w.sum() * E).astype(np.int64)
    sizes[0] += E - sizes.sum()
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    if a.degree == "powerlaw":
        p = 1.0 / np.arange(1, n_per + 1) ** 1.0
        p /= p.sum()
        draw = lambda s: (rng.integers(0, a.graphs, s) * n_per + rng.choice(n_per, s, p=p))
    else:
        draw = lambda s: rng.integers(0, N, s)
    tgt_np = np.concatenate([np.sort(draw(s)) for s in sizes]).astype(np.int32)
    src_np = (tgt_np // n_per * n_per + rng.integers(0, n_per, E)).astype(np.int32)
    tgt, src = torch.from_numpy(tgt_np).cuda(), torch.from_numpy(src_np).cuda()
    tgt_ptr, tgt_msgs = csr(tgt_np, N)
    src_ptr, src_msgs = csr(src_np, N)
    W = torch.randn(T, 2 * Din, Dm, device="cuda") / 16
    gq = torch.randn(N, Dm, device="cuda")
    # a realistic winner table: every (node, channel) won by one of the node's incoming messages
    order = torch.argsort(tgt.long(), stable=True)
    first = torch.searchsorted(tgt.long()[order], torch.arange(N, device="cuda"))
    deg = torch.bincount(tgt.long(), minlength=N)
    pick = (torch.rand(N, Dm, device="cuda") * deg.clamp(min=1)[:, None]).long().clamp(max=E - 1)
    arg = order[(first[:, None] + pick).clamp(max=E - 1)].to(torch.int32)
    arg[deg == 0] = -1
    won = arg[tgt.long()] == torch.arange(E, device="cuda", dtype=torch.int32)[:, None]
    wts = (1 << torch.arange(32, device="cuda", dtype=torch.int64))
    bits = (won.view(E, Dm // 32, 32).long() * wts).sum(-1).to(torch.int32)
    del won, pick, arg
    wt = W.transpose(1, 2).contiguous()
    gqp, wp = ops.pack_bf16x3(gq), ops.pack_weights_x6(W, False)
    lib = ops.load_library()
    g_h = torch.empty(N, Din, device="cuda")

    def scatter(g_a, out):
        ops._check(lib.bl_mp_scatter_grad(g_a.data_ptr(), g_a.stride(0), src_ptr.data_ptr(), src_msgs.data_ptr(), tgt_ptr.data_ptr(),
                                          tgt_msgs.data_ptr(), N, Din, 0, out.data_ptr(), out.stride(0), None, ops._stream()), "scatter")
        return out

    def shipped():
        g_a = ops.gemm_rows_x6([(gqp, tgt, Dm)], wp, E, 2 * Din, group_ptr=ptr, G=T, win_bits=bits)
        return scatter(g_a, g_h)

    def gemm_only():
        return ops.gemm_rows_x6([(gqp, tgt, Dm)], wp, E, 2 * Din, group_ptr=ptr, G=T, win_bits=bits)

    def vec():
        g_a = ops.routed_dgrad_vec(gq, tgt, bits, ptr, T, wt, E, 2 * Din)
        return scatter(g_a, g_h)

    def vec_only():
        return ops.routed_dgrad_vec(gq, tgt, bits, ptr, T, wt, E, 2 * Din)

    out_f = torch.empty(N, Din, device="cuda")

    def fused():
        out_f.zero_()
        ops.routed_dgrad_nodes(gq, src, tgt, bits, ptr, T, wt, E, Din, out_f)
        return out_f

    g_src = torch.empty(E, Din, device="cuda")

    def hybrid():
        out_f.zero_()
        ops.routed_dgrad_nodes(gq, src, tgt, bits, ptr, T, wt, E, Din, out_f, src_rows=g_src)
        ops._check(lib.bl_mp_scatter_grad(g_src.data_ptr(), g_src.stride(0), src_ptr.data_ptr(), src_msgs.data_ptr(), None, None, N, Din, 1,
                                          out_f.data_ptr(), out_f.stride(0), None, ops._stream()), "scatter")
        return out_f

    def hybrid_kernel_only():
        ops.routed_dgrad_nodes(gq, src, tgt, bits, ptr, T, wt, E, Din, out_f, src_rows=g_src)

    want = shipped().clone()
    scale = float(want.abs().max())
    got_v = vec().clone()
    got_f = fused().clone()
    torch.cuda.synchronize()
    print(f"N={N} E={E} Din={Din} Dm={Dm} T={T} degree={a.degree}; largest |g_h| {scale:.3f}")
    print(f"vec + scatter vs shipped: max |diff| {float((got_v - want).abs().max()):.3e}")
    print(f"fused nodes    vs shipped: max |diff| {float((got_f - want).abs().max()):.3e}")
    if Din % 64 == 0 and Din >= 128:
        lo, hi = torch.zeros(N, Din // 2, device="cuda"), torch.zeros(N, Din // 2, device="cuda")
        ops.routed_dgrad_nodes(gq, src, tgt, bits, ptr, T, wt, E, Din, lo, hi)
        print(f"fused split    vs shipped: max |diff| {float((torch.cat([lo, hi], 1) - want).abs().max()):.3e}")
    print(f"(a) shipped routed bf16x6 GEMM + node sums: {timeit(shipped):.3f} ms  (GEMM alone {timeit(gemm_only):.3f})")
    print(f"(b) vector non-zeros -> g_a + node sums:     {timeit(vec):.3f} ms  (kernel alone {timeit(vec_only):.3f})")
    print(f"(c) vector non-zeros, node sums fused:       {timeit(fused):.3f} ms  (incl. zero-fill)")
    got_h = hybrid().clone()
    print(f"hybrid         vs shipped: max |diff| {float((got_h - want).abs().max()):.3e}")
    print(f"(d) target half fused, source half rows + sums: {timeit(hybrid):.3f} ms  (kernel alone {timeit(hybrid_kernel_only):.3f})")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound per-node kernels of one message-passing layer at the BASELINE c2 layer shape:
segmented max + LayerNorm (forward) and the segmented sums of per-message rows (backward).  Reports the time and the rate
of the ALGORITHMIC bytes (what the kernel must read and write once).  `--lib PATH` times another build of the library
(A/B of a kernel change on one box)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np
import torch

from buglab.models import hip_ops as ops


def timeit(f, iters=30):
    for _ in range(5):
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
    ap.add_argument("--dm", type=int, default=128)
    ap.add_argument("--din", type=int, default=128)
    ap.add_argument("--types", type=int, default=16)
    ap.add_argument("--degree", default="uniform")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.lib:
        ops._lib = ops.load_library(a.lib)
    n_per, e_per = 2000, 10000
    N, E, Dm, Din, T = a.graphs * n_per, a.graphs * e_per, a.dm, a.din, a.types
    rng = np.random.default_rng(0)
    w = 1.0 / np.arange(1, T + 1)
    sizes = np.floor(w / w.sum() * E).astype(np.int64)
    sizes[0] += E - sizes.sum()
    if a.degree == "powerlaw":
        p = 1.0 / np.arange(1, n_per + 1) ** 1.0
        p /= p.sum()
        draw = lambda s: (rng.integers(0, a.graphs, s) * n_per + rng.choice(n_per, s, p=p))
    else:
        draw = lambda s: rng.integers(0, N, s)
    tgt = np.concatenate([np.sort(draw(s)) for s in sizes]).astype(np.int32)
    src = (tgt // n_per * n_per + rng.integers(0, n_per, E)).astype(np.int32)
    tgt_ptr, tgt_msgs = csr(tgt, N)
    src_ptr, src_msgs = csr(src, N)
    deg = np.bincount(tgt, minlength=N) + np.bincount(src, minlength=N)
    node_order = torch.from_numpy(np.argsort(-deg, kind="stable").astype(np.int32)).cuda()
    x = torch.randn(E, Dm, device="cuda")
    g, b = torch.ones(Dm, device="cuda"), torch.zeros(Dm, device="cuda")

    def segmax():
        return ops.segment_max(x, tgt_ptr, tgt_msgs, N, act=ops.ACT_GELU, ln=(g, b), want_dact=True, want_bits=True, seg_order=node_order)

    t = timeit(segmax)
    nbytes = E * Dm * 4 + N * Dm * 4 * 4 + E * (Dm // 32) * 4 + E * 4  # messages; out, arg, ln_out, dact; routing bits; item ids
    print(f"segment_max + LayerNorm  N={N} E={E} D={Dm} {a.degree}: {t:.3f} ms   {nbytes / t / 1e6:.0f} GB/s of {nbytes / 1e6:.0f} MB")

    lib = ops.load_library()
    g_a = torch.randn(E, 2 * Din, device="cuda")
    g_h = torch.empty(N, Din, device="cuda")

    def sums_both():
        ops._check(lib.bl_mp_scatter_grad(g_a.data_ptr(), g_a.stride(0), src_ptr.data_ptr(), src_msgs.data_ptr(), tgt_ptr.data_ptr(),
                                          tgt_msgs.data_ptr(), N, Din, 0, g_h.data_ptr(), g_h.stride(0), node_order.data_ptr(), ops._stream()), "scatter")

    g_s = torch.randn(E, Din, device="cuda")

    def sums_src():
        ops._check(lib.bl_mp_scatter_grad(g_s.data_ptr(), g_s.stride(0), src_ptr.data_ptr(), src_msgs.data_ptr(), None, None, N, Din, 1,
                                          g_h.data_ptr(), g_h.stride(0), node_order.data_ptr(), ops._stream()), "scatter")

    t = timeit(sums_both)
    nbytes = E * 2 * Din * 4 + N * Din * 4 + 2 * E * 4
    print(f"node sums, both halves   [E, {2 * Din}] rows: {t:.3f} ms   {nbytes / t / 1e6:.0f} GB/s of {nbytes / 1e6:.0f} MB")
    t = timeit(sums_src)
    nbytes = E * Din * 4 + 2 * N * Din * 4 + E * 4
    print(f"node sums, source halves [E, {Din}] rows (+=): {t:.3f} ms   {nbytes / t / 1e6:.0f} GB/s of {nbytes / 1e6:.0f} MB")


if __name__ == "__main__":
    main()

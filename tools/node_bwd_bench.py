#!/usr/bin/env python
"""Micro-benchmark: the node update's backward chain as one kernel (bl_node_update_bwd) against the three kernels it replaces
(bl_act_bwd_packed -> bl_gemm_rows_x6_epi -> bl_layernorm_bwd) at BASELINE config c2's layer shapes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import math

import torch

from buglab.models import hip_ops as ops
# (BL_HIP_LIB=path selects an experiment build of the library: tools/experiments/node_bwd_variants.sh)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=128000)
    ap.add_argument("--shapes", default="128x128,256x128,256x256", help="Dm x Dout list")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dropout", type=float, default=0.2)
    a = ap.parse_args()
    N = a.nodes
    for shp in a.shapes.split(","):
        Dm, Dout = (int(x) for x in shp.split("x"))
        torch.manual_seed(0)
        g = torch.randn(N, Dout, device="cuda")
        y = torch.tanh(torch.randn(N, Dout, device="cuda"))
        Wd = torch.randn(Dm, Dout, device="cuda") / math.sqrt(Dm)
        agg = torch.randn(N, Dm, device="cuda")
        mean, var = agg.mean(1), agg.var(1, unbiased=False)
        rstd = torch.rsqrt(var + 1e-5)
        gamma = torch.rand(Dm, device="cuda") + 0.5
        dact = torch.rand(N, Dm, device="cuda")
        wnk = ops.pack_weights_x6(Wd.unsqueeze(0), False)
        drop = ops.Dropout(a.dropout, 1, 2) if a.dropout > 0 else ops.NO_DROPOUT
        gb, gg, gbeta = (torch.zeros(Dout, device="cuda"), torch.zeros(Dm, device="cuda"), torch.zeros(Dm, device="cuda"))

        def fused():
            ops.node_update_bwd(g, y, drop, wnk, agg, mean, rstd, gamma, dact, gb, gg, gbeta, want_f32=True)

        def chain():
            gzp = ops.act_bwd_packed(g, y, ops.ACT_TANH, drop, gb)
            gln = ops.gemm_rows_x6([(gzp, None, Dout)], wnk, N, Dm)
            ops.layernorm_bwd(gln, agg, mean, rstd, gamma, gg, gbeta, post_scale=dact, want="both")

        byts_f = N * (Dout * 8 + Dout * 6 + Dm * 8 + Dm * 6 + Dm * 4)
        byts_c = byts_f + N * (Dout * 6 + Dm * 8)
        for name, fn, byts in (("fused", fused, byts_f), ("three kernels", chain, byts_c)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"N={N} Dm={Dm} Dout={Dout} {name:14s}: {ms:.3f} ms  {byts / ms / 1e6:.0f} GB/s of its own algorithmic bytes ({byts / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()

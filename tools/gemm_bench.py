#!/usr/bin/env python
"""Micro-benchmark of the three MFMA GEMM forms at BASELINE config c2 shapes (one MP layer's
message GEMM, input-gradient GEMM and weight-gradient GEMM).  Used to tune csrc/bl_gemm.hip."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np
import torch

from buglab.models import hip_ops as ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=128000)
    ap.add_argument("--msgs", type=int, default=640000)
    ap.add_argument("--din", type=int, default=128)
    ap.add_argument("--dm", type=int, default=128)
    ap.add_argument("--types", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5, help="interleaved measurement rounds per variant")
    ap.add_argument("--which", default="fwd,nk,wgrad")
    ap.add_argument("--order", default="type", help="type: type-major; chunk: (graph chunk, type)-major")
    ap.add_argument("--chunk", type=int, default=1, help="graphs per chunk")
    ap.add_argument("--mixed", type=float, default=0.0,
                    help="> 0: every timed launch is followed by an HBM-bound filler (a device copy of that many GB) and only the GEMM's own "
                         "events are summed.  Back to back, a GEMM holds the package at its 1 400 W limit and the clock at 1.8 GHz; in the "
                         "training step the kernels alternate and the clock is ~2.2 GHz (tools/experiments/step_power.sh): this mode times "
                         "a variant under THOSE conditions")
    ap.add_argument("--kcaps", default="", help="comma list: also time wgrad_x6 with these row caps per workgroup (bl_set_wgrad_kchunk_cap)")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    N, E, Din, Dm, T = a.nodes, a.msgs, a.din, a.dm, a.types
    w = 1.0 / np.arange(1, T + 1)
    sizes = np.floor(w / w.sum() * E).astype(np.int64)
    sizes[0] += E - sizes.sum()
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    # graph-local sources like the real collator: messages of a type sorted by target
    tgt = np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes]).astype(np.int32)
    src = (tgt // 2000 * 2000 + rng.integers(0, 2000, E)).clip(0, N - 1).astype(np.int32)
    gw_t = None
    if a.order == "chunk":
        typ = np.repeat(np.arange(T), sizes)
        chunk = tgt // (2000 * a.chunk)
        nchunk = int(chunk.max()) + 1
        order = np.lexsort((tgt, typ, chunk))
        src, tgt, typ, chunk = src[order], tgt[order], typ[order], chunk[order]
        key = chunk * T + typ
        counts = np.bincount(key, minlength=nchunk * T)
        ptr = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32).cuda()
        gw_t = torch.tensor(np.tile(np.arange(T), nchunk), dtype=torch.int32).cuda()
        T_groups = nchunk * T

        def tile_map(piece):
            gp = np.concatenate([[0], np.cumsum(counts)])
            nt = (counts + piece - 1) // piece
            grp = np.repeat(np.arange(len(counts)), nt)
            first = np.arange(nt.sum()) - np.repeat(np.cumsum(nt) - nt, nt)
            return torch.tensor(np.stack([grp, gp[grp] + first * piece], 1), dtype=torch.int32).cuda()

        tm128, tm1024 = tile_map(128), tile_map(1024)
    else:
        T_groups = T
        tm128 = tm1024 = None
    h = torch.randn(N, Din, device="cuda")
    W = torch.randn(T, 2 * Din, Dm, device="cuda") / np.sqrt(2 * Din)
    G = torch.randn(E, Dm, device="cuda")
    src, tgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    out = torch.empty(E, Dm, device="cuda")
    ga = torch.empty(E, 2 * Din, device="cuda")
    gw = torch.zeros_like(W)
    flop = 2.0 * E * 2 * Din * Dm
    fns = {
        "fwd": lambda: ops.gemm_rows([(h, src), (h, tgt)], W, E, Dm, b_group_stride=2 * Din * Dm, ldb=Dm, group_ptr=ptr, G=T, out=out),
        "nk": lambda: ops.gemm_rows([(G, None)], W, E, 2 * Din, b_is_nk=True, b_group_stride=2 * Din * Dm, ldb=Dm, group_ptr=ptr, G=T, out=ga),
        "wgrad": lambda: ops.gemm_wgrad([(h, src), (h, tgt)], G, E, Dm, gw, gw_group_stride=2 * Din * Dm, group_ptr=ptr, G=T),
    }
    hp, wtp, wp = ops.pack_bf16x3(h), ops.pack_weights_x6(W, True), ops.pack_weights_x6(W, False)
    gq = torch.randn(N, Dm, device="cuda")
    gqp = ops.pack_bf16x3(gq)
    arg = torch.randint(0, E, (N, Dm), device="cuda", dtype=torch.int32)
    fns["fwd_x6"] = lambda: ops.gemm_rows_x6([(hp, src, Din), (hp, tgt, Din)], wtp, E, Dm, group_ptr=ptr, G=T_groups, group_w=gw_t)
    # a realistic winner table: every (node, channel) won by one of the node's incoming messages
    order = torch.argsort(tgt.long(), stable=True)
    first = torch.searchsorted(tgt.long()[order], torch.arange(N, device="cuda"))
    deg = torch.bincount(tgt.long(), minlength=N)
    pick = (torch.rand(N, Dm, device="cuda") * deg.clamp(min=1)[:, None]).long().clamp(max=E - 1)
    arg_real = order[(first[:, None] + pick).clamp(max=E - 1)].to(torch.int32)
    arg_real[deg == 0] = -1
    won = arg_real[tgt.long()] == torch.arange(E, device="cuda", dtype=torch.int32)[:, None]
    wts = (1 << torch.arange(32, device="cuda", dtype=torch.int64))
    bits_real = (won.view(E, Dm // 32, 32).long() * wts).sum(-1).to(torch.int32)   # low 32 bits, two's complement
    gw6 = torch.zeros_like(W)
    fns["wgrad_routed"] = lambda: ops.gemm_wgrad_routed([(h, src), (h, tgt)], gq, tgt, arg_real, E, Dm, gw, gw_group_stride=2 * Din * Dm, group_ptr=ptr, G=T)
    fns["wgrad_x6"] = lambda: ops.gemm_wgrad_routed_x6([(hp, src, Din), (hp, tgt, Din)], gqp, tgt, bits_real, E, Dm, gw6, gw_group_stride=2 * Din * Dm, group_ptr=ptr, G=T_groups, group_w=gw_t)
    def wgrad_x6_tile128():
        prev = ops.set_wgrad_tile(128)
        try:
            fns["wgrad_x6"]()
        finally:
            ops.set_wgrad_tile(prev)

    fns["wgrad_x6_t128"] = wgrad_x6_tile128  # A/B: the 128 x 128 tile (wgrad_x6 = the default, wide where it applies)
    def with_cap(cap):
        def run():
            prev = ops.set_wgrad_kchunk_cap(cap)
            try:
                fns["wgrad_x6"]()
            finally:
                ops.set_wgrad_kchunk_cap(prev)
        return run

    caps = [int(c) for c in a.kcaps.split(",") if c]
    for cap in caps:
        fns[f"wgrad_x6_cap{cap}"] = with_cap(cap)
    if caps:  # same sums whatever the chunking (up to the order of the fp32 additions)
        gw6.zero_(); fns["wgrad_x6"](); ref = gw6.clone()
        for cap in caps:
            gw6.zero_(); fns[f"wgrad_x6_cap{cap}"]()
            err = ((gw6 - ref).abs().max() / ref.abs().max()).item()
            print(f"cap {cap}: max |diff| / max |ref| = {err:.2e}", flush=True)
            assert err < 1e-5
    fns["nk_x6"] = lambda: ops.gemm_rows_x6([(gqp, tgt, Dm)], wp, E, 2 * Din, group_ptr=ptr, G=T_groups, group_w=gw_t, win_bits=bits_real)
    if ops.rows_x6w_ok(Dm, 2 * Din) and a.order == "type":  # the wide (128 x 256 tile, LDS-DMA) form of the forward GEMM: bit-identical
        wtp_w = ops.pack_weights_x6w(W, True)
        fns["fwd_x6w"] = lambda: ops.gemm_rows_x6([(hp, src, Din), (hp, tgt, Din)], wtp_w, E, Dm, group_ptr=ptr, G=T, wide=True)
        same = torch.equal(fns["fwd_x6w"](), fns["fwd_x6"]())
        print(f"fwd_x6w == fwd_x6 bit for bit: {same}", flush=True)
        assert same
    if ops.rows_x6w_ok(2 * Din, Dm) and a.order == "type":  # ... and of the routed input-gradient GEMM
        wp_w = ops.pack_weights_x6w(W, False)
        fns["nk_x6w"] = lambda: ops.gemm_rows_x6([(gqp, tgt, Dm)], wp_w, E, 2 * Din, group_ptr=ptr, G=T, win_bits=bits_real, wide=True)
        same = torch.equal(fns["nk_x6w"](), fns["nk_x6"]())
        print(f"nk_x6w == nk_x6 bit for bit: {same}", flush=True)
        assert same
    # f16x3 (csrc/bl_gemm_h3.hip): two fp16 planes per operand, three MFMA terms
    hp16, wtp16, wp16 = ops.pack_f16x2(h, ops.H3_ROW_SCALE), ops.pack_weights_h3(W, True), ops.pack_weights_h3(W, False)
    gq_am = ops.amax(gq)
    gqp16 = ops.pack_f16x2(gq, 1.0, amax=gq_am)
    gw16 = torch.zeros_like(W)
    fns["fwd_h3"] = lambda: ops.gemm_rows_h3([(hp16, src, Din), (hp16, tgt, Din)], wtp16, E, Dm, out_scale=1.0 / (ops.H3_ROW_SCALE * ops.H3_W_SCALE),
                                             group_ptr=ptr, G=T_groups, group_w=gw_t)
    fns["nk_h3"] = lambda: ops.gemm_rows_h3([(gqp16, tgt, Dm)], wp16, E, 2 * Din, out_scale=1.0 / ops.H3_W_SCALE, a_amax=gq_am, group_ptr=ptr,
                                            G=T_groups, group_w=gw_t, win_bits=bits_real)
    fns["wgrad_h3"] = lambda: ops.gemm_wgrad_h3([(hp16, src, Din), (hp16, tgt, Din)], gqp16, E, Dm, gw16, out_scale=1.0 / ops.H3_ROW_SCALE, g_idx=tgt,
                                                win_bits=bits_real, g_amax=gq_am, gw_group_stride=2 * Din * Dm, group_ptr=ptr, G=T_groups, group_w=gw_t)
    fns["pack_h16"] = lambda: ops.pack_f16x2(h, ops.H3_ROW_SCALE)
    fns["pack_h"] = lambda: ops.pack_bf16x3(h)
    fns["pack_wt"] = lambda: ops.pack_weights_x6(W, True)
    names = [n for n in a.which.split(",") if n in fns or print(f"(skipping {n}: shape not supported)")] + [f"wgrad_x6_cap{cap}" for cap in caps]
    times = {n: [] for n in names}
    filler_src = torch.empty(int(a.mixed * 2 ** 30) // 4, device="cuda") if a.mixed > 0 else None
    filler_dst = torch.empty_like(filler_src) if a.mixed > 0 else None
    for rnd in range(a.rounds):  # interleaved rounds in ONE process: report median and min per variant
        for name in names:
            f = fns[name]
            for _ in range(2 if rnd == 0 else 1):
                f()
            torch.cuda.synchronize()
            if a.mixed > 0:  # GEMM, filler, GEMM, filler, ...: sum of the GEMM's own event spans
                for _ in range(10):  # let the power controller settle on the mix
                    f(); filler_dst.copy_(filler_src)
                evs = []
                for _ in range(a.iters):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); f(); e1.record()
                    filler_dst.copy_(filler_src)
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                times[name].append(sum(x.elapsed_time(y) for x, y in evs) / a.iters)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / a.iters)
    for name in names:
        ts = sorted(times[name])
        ms, mn = ts[len(ts) // 2], ts[0]
        print(f"{name:14s} E={E} Din={Din} Dm={Dm}: median {ms:.3f} ms (min {mn:.3f})  {flop / ms / 1e9:.1f} TFLOP/s  "
              f"({flop / ms / 1e9 / 416.7:.1%} of the bf16x6 peak, {flop / ms / 1e9 / 157.3:.1%} of fp32 MFMA peak)", flush=True)


if __name__ == "__main__":
    main()

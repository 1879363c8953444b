"""GPU: every C-ABI entry point of libbuglab_hip against a plain PyTorch (fp64 where cheap)
reference of the same op, on seeded inputs chosen to hit the edge cases: empty groups/segments,
partial tiles, widths that are not multiples of the tile, hub segments longer than one wave batch,
multiple gathered sources.  Tolerances are written next to each check (fp32 accumulate-order noise)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import buglab_oracle as O


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from buglab.models import hip_ops

    hip_ops.load_library()
    return hip_ops


def _dev(a, dtype=None):
    t = torch.as_tensor(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def _groups(rng, sizes):
    ptr = np.zeros(len(sizes) + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(sizes)
    return ptr


@pytest.mark.parametrize("Din,Dm,sizes", [(64, 96, [130, 0, 1, 257, 64]), (32, 128, [5, 300]), (128, 256, [129, 128, 127]), (24, 40, [70, 3])])
def test_gemm_rows_grouped_gathered(ops, Din, Dm, sizes):
    rng = np.random.default_rng(0)
    N, T, E = 211, len(sizes), int(sum(sizes))
    h = torch.randn(N, Din, dtype=torch.float32)
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    src = rng.integers(0, N, E).astype(np.int32)
    tgt = rng.integers(0, N, E).astype(np.int32)
    ptr = _groups(rng, sizes)
    ref = torch.zeros(E, Dm, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        a = torch.cat([h[src[lo:hi]], h[tgt[lo:hi]]], -1).double()
        ref[lo:hi] = a @ W[t].double()
    hd, Wd = _dev(h), _dev(W)
    out = ops.gemm_rows([(hd, _dev(src)), (hd, _dev(tgt))], Wd, E, Dm, b_group_stride=2 * Din * Dm, ldb=Dm,
                        group_ptr=_dev(ptr), G=T)
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max() < 2e-5  # K <= 256 products of O(1) values

    # transposed-B form (input gradient): dA = G . W_t^T
    G = torch.randn(E, Dm)
    ref2 = torch.zeros(E, 2 * Din, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        ref2[lo:hi] = G[lo:hi].double() @ W[t].double().T
    out2 = ops.gemm_rows([(_dev(G), None)], Wd, E, 2 * Din, b_is_nk=True, b_group_stride=2 * Din * Dm, ldb=Dm,
                         group_ptr=_dev(ptr), G=T)
    assert (out2.cpu().double() - ref2).abs().max() < 2e-5

    # weight gradient: gW_t = A_t^T G_t (fp32 atomics across row chunks)
    gw = torch.zeros(T, 2 * Din, Dm, device="cuda")
    ops.gemm_wgrad([(hd, _dev(src)), (hd, _dev(tgt))], _dev(G), E, Dm, gw, gw_group_stride=2 * Din * Dm, group_ptr=_dev(ptr), G=T)
    ref3 = torch.zeros(T, 2 * Din, Dm, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        a = torch.cat([h[src[lo:hi]], h[tgt[lo:hi]]], -1).double()
        ref3[t] = a.T @ G[lo:hi].double()
    scale = max(1.0, float(ref3.abs().max()))
    assert (gw.cpu().double() - ref3).abs().max() < 3e-5 * scale


@pytest.mark.parametrize("act", ["none", "relu", "sigmoid", "tanh"])
def test_gemm_rows_bias_act_dropout_three_sources(ops, act):
    rng = np.random.default_rng(1)
    R, H = 333, 48
    a, b, c = torch.randn(50, H), torch.randn(70, H), torch.randn(20, H)
    ia, ib, ic = (rng.integers(0, n, R).astype(np.int32) for n in (50, 70, 20))
    W = torch.randn(3 * H, 52) / math.sqrt(3 * H)
    bias = torch.randn(52)
    x = torch.cat([a[ia], b[ib], c[ic]], -1).double()
    z = x @ W.double() + bias.double()
    ref = {"none": z, "relu": torch.relu(z), "sigmoid": torch.sigmoid(z), "tanh": torch.tanh(z)}[act]
    drop = ops.Dropout(0.3, 1234, 7)
    keep = torch.from_numpy(O.dropout_keep_mask(1234, 7, R * 52, 0.3)).view(R, 52)
    ref_d = ref * keep / (1 - np.float32(0.3)).astype(np.float64)
    out = ops.gemm_rows([(_dev(a), _dev(ia)), (_dev(b), _dev(ib)), (_dev(c), _dev(ic))], _dev(W), R, 52, bias=_dev(bias),
                        act=ops._ACTS[act], drop=drop)
    assert (out.cpu().double() - ref_d).abs().max() < 3e-5
    assert ((out.cpu() == 0) == (~keep | (ref_d == 0))).all()  # identical mask, bit for bit


def _segmax_ref_values(x, act):
    """what the max compares, and what is applied to the aggregate afterwards (gelu_aggregated = ptgnn's order: gelu(max x))"""
    before = O._gelu(x) if act == "gelu" else x
    after = O._gelu if act == "gelu_aggregated" else (lambda v: v)
    return before, after


@pytest.mark.parametrize("D,act", [(96, "gelu"), (320, "none"), (128, "gelu"), (8, "none"), (128, "gelu_aggregated"), (96, "gelu_aggregated"),
                                   (512, "gelu_aggregated")])
def test_segment_max_layernorm_fwd_bwd(ops, D, act):
    rng = np.random.default_rng(2)
    nseg, E = 57, 900
    seg = rng.integers(0, nseg, E)
    seg[seg == 5] = 6  # empty segment
    seg[:200] = 11  # hub: more than one 64-item batch
    order = np.argsort(seg, kind="stable").astype(np.int32)
    ptr = np.zeros(nseg + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(np.bincount(seg, minlength=nseg))
    x = torch.randn(E, D)
    x[3] = x[1]  # exact tie inside a segment? force same segment
    seg[3] = seg[1]
    x[150] = x[1]  # and one more copy far down the hub segment: with the multi-wave hub kernel (seg_order given, below) the
    # tie spans two waves' shares and the FIRST item must still win
    order = np.argsort(seg, kind="stable").astype(np.int32)
    ptr[1:] = np.cumsum(np.bincount(seg, minlength=nseg))
    g, b = torch.randn(D), torch.randn(D)
    xa, after = _segmax_ref_values(x.double(), act)
    ref, arg = O.scatter_max_with_arg(xa, torch.from_numpy(seg), nseg)
    ref = after(ref)
    ref_ln = torch.nn.functional.layer_norm(ref, (D,), g.double(), b.double(), eps=1e-5)
    out, a, ln_out, mean, rstd, dact, wbits = ops.segment_max(_dev(x), _dev(ptr), _dev(order), nseg, act=ops._ACTS[act], ln=(_dev(g), _dev(b)),
                                                              want_dact=True, want_bits=True)
    assert (out.cpu().double() - ref).abs().max() < 1e-6
    a_ref = torch.where(arg == E, torch.full_like(arg, -1), arg)
    assert (a.cpu().long() == a_ref).all()
    assert (ln_out.cpu().double() - ref_ln).abs().max() < 2e-5
    # per-item routing bitmask: bit d of item i <=> i is the arg-max of channel d of its segment
    won = (a_ref[torch.from_numpy(seg).long()] == torch.arange(E)[:, None]).numpy()
    W32 = (D + 31) // 32
    wpad = np.zeros((E, W32 * 32), dtype=bool)
    wpad[:, :D] = won
    ref_bits = np.packbits(wpad.reshape(E, W32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(E, W32)
    assert (wbits.cpu().numpy().view(np.uint32) == ref_bits).all()
    # a processing order (hubs first in the collator) changes nothing
    perm = _dev(np.random.default_rng(5).permutation(nseg).astype(np.int32))
    o2 = ops.segment_max(_dev(x), _dev(ptr), _dev(order), nseg, act=ops._ACTS[act], ln=(_dev(g), _dev(b)), want_dact=True, want_bits=True, seg_order=perm)
    for t1, t2 in zip((out, a, ln_out, mean, rstd, dact, wbits), o2):
        assert torch.equal(t1, t2)
    # backward of the max (gather form) against autograd through the oracle
    if D % 4 == 0:
        go = torch.randn(nseg, D)
        xr = x.double().requires_grad_(True)
        xa, after = _segmax_ref_values(xr, act)
        after(O.scatter_max_with_arg(xa, torch.from_numpy(seg), nseg)[0]).backward(go.double())
        gx = ops.segment_max_bwd(_dev(go), a, _dev(x), _dev(seg.astype(np.int32)), act=ops._ACTS[act])
        assert (gx.cpu().double() - xr.grad).abs().max() < 1e-5
        # routed GEMMs: the same gradient, never materialised: G[i,:] = (go * dact)[seg[i]] masked to i's wins
        gq = (_dev(go) * dact).contiguous()
        W = torch.randn(3, 40, D) / math.sqrt(D)          # [groups, N_out, K=D] used transposed
        gptr = np.array([0, 300, 300, E], dtype=np.int32)
        dA = ops.gemm_rows_routed(gq, _dev(seg.astype(np.int32)), a, _dev(W), E, 40, b_group_stride=40 * D, ldb=D, group_ptr=_dev(gptr), G=3)
        ref_dA = torch.zeros(E, 40, dtype=torch.float64)
        for t in range(3):
            lo, hi = gptr[t], gptr[t + 1]
            ref_dA[lo:hi] = xr.grad[lo:hi] @ W[t].double().T
        assert (dA.cpu().double() - ref_dA).abs().max() < 3e-5 * max(1.0, float(ref_dA.abs().max()))
        hfeat = torch.randn(77, 24)
        hidx = np.random.default_rng(9).integers(0, 77, E).astype(np.int32)
        gw = torch.zeros(3, 24, D, device="cuda")
        ops.gemm_wgrad_routed([(_dev(hfeat), _dev(hidx))], gq, _dev(seg.astype(np.int32)), a, E, D, gw, gw_group_stride=24 * D, group_ptr=_dev(gptr), G=3)
        ref_gw = torch.zeros(3, 24, D, dtype=torch.float64)
        for t in range(3):
            lo, hi = gptr[t], gptr[t + 1]
            ref_gw[t] = hfeat[hidx[lo:hi]].double().T @ xr.grad[lo:hi]
        assert (gw.cpu().double() - ref_gw).abs().max() < 3e-5 * max(1.0, float(ref_gw.abs().max()))
        # LayerNorm backward
        gy = torch.randn(nseg, D)
        rr = ref.clone().float().requires_grad_(True)
        gg, bb = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        torch.nn.functional.layer_norm(rr, (D,), gg, bb, eps=1e-5).backward(gy)
        d_g, d_b = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        gxx = ops.layernorm_bwd(_dev(gy), out, mean, rstd, _dev(g), d_g, d_b)
        # rows with zero variance (the empty segment) have rstd = 1/sqrt(eps) ~ 316: compare relative to the largest entry
        assert (gxx.cpu() - rr.grad).abs().max() < 5e-5 * max(1.0, float(rr.grad.abs().max()))
        assert (d_g.cpu() - gg.grad).abs().max() < 1e-4 * max(1.0, float(gg.grad.abs().max()))
        assert (d_b.cpu() - bb.grad).abs().max() < 1e-4 * max(1.0, float(bb.grad.abs().max()))
        if D % 8 == 0:  # the same result, emitted bf16x3-packed for the bf16x6 GEMMs
            gx2, gxp = ops.layernorm_bwd(_dev(gy), out, mean, rstd, _dev(g), torch.zeros_like(d_g), torch.zeros_like(d_b), want="both")
            assert torch.equal(gx2, gxx)
            planes = torch.from_numpy((gxp.cpu().numpy().view(np.uint16).astype(np.uint32) << 16).view(np.float32)).view(nseg, 3, D)
            assert (planes.double().sum(1) - gxx.cpu().double()).abs().max() <= 2.0 ** -23 * float(gxx.abs().max())
            assert torch.equal(gxp, ops.pack_bf16x3(gxx))


def test_act_bwd_and_bias(ops):
    R, N = 301, 72
    y_pre = torch.randn(R, N)
    drop = ops.Dropout(0.2, 99, 3)
    keep = torch.from_numpy(O.dropout_keep_mask(99, 3, R * N, 0.2)).view(R, N)
    t = torch.tanh(y_pre.double())
    y = (t * keep / (1 - 0.2)).float()
    g = torch.randn(R, N)
    ref = g.double() * keep / (1 - 0.2) * (1 - t * t)
    gb = torch.zeros(N, device="cuda")
    gz = ops.act_bwd(_dev(g), _dev(y), ops.ACT_TANH, drop, gb)
    assert (gz.cpu().double() - ref).abs().max() < 1e-5
    assert (gb.cpu().double() - ref.sum(0)).abs().max() < 1e-4


def test_mp_scatter_grad(ops):
    rng = np.random.default_rng(3)
    N, E, Din = 90, 700, 96
    src, tgt = rng.integers(0, N, E), rng.integers(0, N, E)
    tgt[:150] = 7
    ga = torch.randn(E, 2 * Din)
    ref = torch.zeros(N, Din, dtype=torch.float64)
    ref.index_add_(0, torch.from_numpy(src), ga[:, :Din].double())
    ref.index_add_(0, torch.from_numpy(tgt), ga[:, Din:].double())
    from buglab.data.collate import _csr

    sp, sm = _csr(src, N)
    tp, tm = _csr(tgt, N)
    gh = torch.empty(N, Din, device="cuda")
    lib = ops.load_library()
    d_ga, d_sp, d_sm, d_tp, d_tm = (_dev(a) for a in (ga, sp, sm, tp, tm))  # keep the device copies alive
    ops._check(lib.bl_mp_scatter_grad(d_ga.data_ptr(), 2 * Din, d_sp.data_ptr(), d_sm.data_ptr(), d_tp.data_ptr(),
                                      d_tm.data_ptr(), N, Din, 0, gh.data_ptr(), Din, None, torch.cuda.current_stream().cuda_stream), "scatter")
    torch.cuda.synchronize()
    assert (gh.cpu().double() - ref).abs().max() < 1e-4
    # any processing order (the collator puts hubs first) gives the same rows
    order = _dev(np.random.default_rng(1).permutation(N).astype(np.int32))
    gh2 = torch.full_like(gh, float("nan"))
    ops._check(lib.bl_mp_scatter_grad(d_ga.data_ptr(), 2 * Din, d_sp.data_ptr(), d_sm.data_ptr(), d_tp.data_ptr(),
                                      d_tm.data_ptr(), N, Din, 0, gh2.data_ptr(), Din, order.data_ptr(), torch.cuda.current_stream().cuda_stream), "scatter")
    torch.cuda.synchronize()
    assert torch.equal(gh2, gh)


def test_segment_log_softmax_fwd_bwd(ops):
    rng = np.random.default_rng(4)
    K, G = 500, 40
    idx = rng.integers(0, G, K)
    idx[idx == 9] = 10
    idx[:130] = 3
    x = (torch.randn(K) * 3).requires_grad_(True)
    ref = O.scatter_log_softmax(x, torch.from_numpy(idx))
    gy = torch.randn(K)
    ref.backward(gy)
    from buglab.data.collate import segments_from_index

    ptr, items = segments_from_index(idx, G)
    xd = _dev(x.detach()).requires_grad_(True)
    y = ops.segment_log_softmax(xd, _dev(ptr), _dev(items), G)
    y.backward(_dev(gy))
    assert (y.detach().cpu() - ref.detach()).abs().max() < 2e-6
    assert (xd.grad.cpu() - x.grad).abs().max() < 1e-5


def test_rowdot_and_scatter_add(ops):
    R, H = 77, 96
    x, w, b = torch.randn(R, H), torch.randn(H), torch.randn(1)
    xd, wd, bd = (_dev(t).requires_grad_(True) for t in (x, w, b))
    y = ops.rowdot(xd, wd, bd)
    gy = torch.randn(R)
    y.backward(_dev(gy))
    assert (y.detach().cpu().double() - (x.double() @ w.double() + b.double())).abs().max() < 1e-5
    assert (xd.grad.cpu() - gy[:, None] * w[None]).abs().max() < 1e-6
    assert (wd.grad.cpu().double() - (gy.double()[:, None] * x.double()).sum(0)).abs().max() < 1e-4
    assert abs(float(bd.grad) - float(gy.sum())) < 1e-4
    idx = np.random.default_rng(5).integers(0, 20, R).astype(np.int32)
    out = torch.zeros(20, 32, device="cuda")
    ops.scatter_add_rows(_dev(x), 16, 32, _dev(idx), out)
    ref = torch.zeros(20, 32, dtype=torch.float64).index_add_(0, torch.from_numpy(idx).long(), x[:, 16:48].double())
    assert (out.cpu().double() - ref).abs().max() < 1e-5


@pytest.mark.parametrize("combination", ["max", "sum", "mean"])
@pytest.mark.parametrize("placement", ["after_pooling", "before_pooling"])
def test_embed_fwd_bwd(ops, placement, combination):
    """subtoken embedder, dropout on the pooled rows (default) or on the embedded subtokens before the pooling;
    max (the registry's default), sum or mean over the subtokens (node_representations["subtoken_combination"])"""
    before = placement == "before_pooling"
    rng = np.random.default_rng(6)
    V, H, N, S = 97, 64, 200, 6
    table = torch.randn(V, H)
    ids = rng.integers(0, V, (N, S)).astype(np.int32)
    lens = rng.integers(1, S + 1, N).astype(np.int32)
    tr = table.clone().requires_grad_(True)
    ref = O.embed_nodes(tr, ids, lens, 0.25, 42, placement, combination)
    go = torch.randn(N, H)
    ref.backward(go)
    td = _dev(table).requires_grad_(True)
    out = ops.embed_subtoken_max(td, _dev(ids), _dev(lens), ops.Dropout(0.25, 42, 0), dropout_before_pooling=before, combination=combination)
    out.backward(_dev(go))
    assert (out.detach().cpu() - ref.detach()).abs().max() < (1e-6 if combination == "max" else 1e-5)
    assert (td.grad.cpu() - tr.grad).abs().max() < 1e-4
    # token-sorted form of the gradient (collator-built occurrence chunks; a hot token spans several chunks)
    from buglab.data.collate import token_occurrence_chunks

    ids[: N // 2, 0] = 7
    tr2 = table.clone().requires_grad_(True)
    O.embed_nodes(tr2, ids, lens, 0.25, 42, placement, combination).backward(go)
    occ, cptr, ctok = token_occurrence_chunks(ids, lens, chunk=16)
    assert (np.diff(cptr) <= 16).all() and cptr[-1] == occ.size == int(lens.sum())
    td2 = _dev(table).requires_grad_(True)
    ops.embed_subtoken_max(td2, _dev(ids), _dev(lens), ops.Dropout(0.25, 42, 0), (_dev(occ), _dev(cptr), _dev(ctok)),
                           dropout_before_pooling=before, combination=combination).backward(_dev(go))
    assert (td2.grad.cpu() - tr2.grad).abs().max() < 1e-4
    # deterministic mode: the serial column walk gives the same gradient
    ops.set_deterministic(True)
    try:
        td3 = _dev(table).requires_grad_(True)
        ops.embed_subtoken_max(td3, _dev(ids), _dev(lens), ops.Dropout(0.25, 42, 0), dropout_before_pooling=before,
                               combination=combination).backward(_dev(go))
        assert (td3.grad.cpu() - tr2.grad).abs().max() < 1e-4
    finally:
        ops.set_deterministic(False)


def test_flat_adam_matches_oracle(ops):
    from buglab.runtime.optim import FlatAdam

    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(13, 7, device="cuda")), torch.nn.Parameter(torch.randn(5, device="cuda"))]
    ref_p = {i: p.detach().cpu().clone() for i, p in enumerate(ps)}
    m = {i: torch.zeros_like(v) for i, v in ref_p.items()}
    v = {i: torch.zeros_like(vv) for i, vv in ref_p.items()}
    opt = FlatAdam(ps, lr=1e-2, clip_gradient_norm=0.5, num_warmup_steps=3)
    for step in range(1, 6):
        opt.zero_grad()
        grads = {i: torch.randn_like(ref_p[i]) * (3.0 if step % 2 else 0.01) for i in ref_p}
        for i, p in enumerate(ps):
            p.grad.add_(grads[i].cuda())
        opt.step()
        O.adam_clip_step(ref_p, grads, m, v, step, lr=1e-2, clip=0.5, warmup=3)
        for i, p in enumerate(ps):
            assert (p.detach().cpu() - ref_p[i]).abs().max() < 2e-6, (step, i)


def test_data_parallel_adam_clips_the_mean_gradient_and_norm_is_reproducible(ops):
    """bl_adam_clip_step_dp gets the SUM over ranks of (graphs x gradient) and the global graph count on the device: it
    must clip and step exactly like one process with the mean gradient -- in both the clipped (big gradient) and the
    unclipped (tiny gradient) regime -- and leave everything alone when the count is 0.  The norm it clips by is
    summed in a fixed order: repeated calls give the same bits (replicas must not drift apart)."""
    torch.manual_seed(1)
    n = 100_003
    p0 = torch.randn(n)
    for count, gscale in ((8.0, 3.0), (8.0, 1e-4)):
        mean_grad = torch.randn(n) * gscale
        ref_p, m, v = {0: p0.clone()}, {0: torch.zeros(n)}, {0: torch.zeros(n)}
        O.adam_clip_step(ref_p, {0: mean_grad}, m, v, 1, lr=1e-2, clip=0.5, warmup=0)
        p, summed = _dev(p0.clone()), _dev(mean_grad * count)
        dm, dv, sq = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
        tail = torch.tensor([count, 1.0, 0.0, 0.0], device="cuda")
        ops.sqnorm(summed, sq)
        first = sq.clone()
        for _ in range(5):
            ops.sqnorm(summed, sq)
            assert torch.equal(sq, first)
        assert abs(float(sq) - float((summed.double() ** 2).sum())) <= 1e-5 * float(sq)
        ops.adam_clip_step_dp(p, summed, dm, dv, sq, tail, clip=0.5, lr=1e-2, step=1)
        assert (p.cpu() - ref_p[0]).abs().max() < 2e-6
        assert (dm.cpu() - m[0]).abs().max() < 1e-6 * max(1.0, gscale)
        before = p.clone()
        ops.adam_clip_step_dp(p, summed, dm, dv, sq, torch.zeros(4, device="cuda"), clip=0.5, lr=1e-2, step=2)
        assert torch.equal(p, before)


@pytest.mark.parametrize("Din,Dm,sizes", [(64, 96, [130, 0, 1, 257, 64]), (32, 128, [5, 300]), (128, 256, [129, 128, 127])])
def test_gemm_rows_bf16x6_is_fp32_accurate(ops, Din, Dm, sizes):
    """fp32-accurate GEMM on the bf16 matrix cores: split hi/mid/lo + six MFMA terms.  The error vs an
    fp64 reference must be in the same class as the exact-fp32 MFMA kernel's (K <= 256: < 2e-5 abs on
    O(1) results; typically ~1e-6)."""
    rng = np.random.default_rng(0)
    N, T, E = 211, len(sizes), int(sum(sizes))
    h = torch.randn(N, Din)
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    src, tgt = rng.integers(0, N, E).astype(np.int32), rng.integers(0, N, E).astype(np.int32)
    ptr = _groups(rng, sizes)
    ref = torch.zeros(E, Dm, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        ref[lo:hi] = torch.cat([h[src[lo:hi]], h[tgt[lo:hi]]], -1).double() @ W[t].double()
    # the packing itself: hi + mid + lo reproduces the fp32 value to ~2^-25
    hp = ops.pack_bf16x3(_dev(h))
    planes = hp.cpu().view(N, 3, Din // 8, 8).permute(0, 2, 1, 3).to(torch.int32).bitwise_and(0xFFFF)
    vals = (planes << 16).view(torch.float32) if False else torch.from_numpy((planes.numpy().astype(np.uint32) << 16).view(np.float32))
    recon = vals[:, :, 0, :] .double() + vals[:, :, 1, :].double() + vals[:, :, 2, :].double()
    assert (recon.reshape(N, Din) - h.double()).abs().max() <= 2.0 ** -24 * float(h.abs().max())
    out = ops.gemm_rows_x6([(hp, _dev(src), Din), (hp, _dev(tgt), Din)], ops.pack_weights_x6(_dev(W), True), E, Dm,
                           group_ptr=_dev(ptr), G=T)
    err = (out.cpu().double() - ref).abs().max()
    exact = ops.gemm_rows([(_dev(h), _dev(src)), (_dev(h), _dev(tgt))], _dev(W), E, Dm, b_group_stride=2 * Din * Dm, ldb=Dm, group_ptr=_dev(ptr), G=T)
    err32 = (exact.cpu().double() - ref).abs().max()
    assert err < 2e-5 and err < 4 * err32 + 1e-6, (float(err), float(err32))


def _unpack_f16x2(p, D):
    """fp64 value of a bl_pack_f16x2 row-packed int16 tensor [R, 2 D]: hi + lo"""
    h = torch.from_numpy(p.cpu().numpy().view(np.float16).astype(np.float64)).view(p.shape[0], 2, D)
    return h.sum(1)


@pytest.mark.parametrize("Din,Dm,sizes", [(64, 96, [130, 0, 1, 257, 64]), (32, 128, [5, 300]), (128, 256, [129, 128, 127]), (128, 128, [3000, 1, 900])])
def test_gemm_rows_f16x3_is_fp32_accurate(ops, Din, Dm, sizes):
    """fp32-accurate GEMM on the fp16 matrix cores (csrc/bl_gemm_h3.hip): two fp16 planes per operand with a power-of-two tensor
    scale, three MFMA terms.  The packing reproduces the fp32 values to 2^-24 (relative, where both planes are normal) or
    2^-25 / scale (absolute); the product's error against fp64 is in the class of the exact-fp32 MFMA kernel's."""
    rng = np.random.default_rng(0)
    N, T, E = 211, len(sizes), int(sum(sizes))
    h = torch.tanh(torch.randn(N, Din)) * 1.25  # a layer input: tanh x dropout scale
    h[0, :5] = torch.tensor([0.0, 1e-9, -3e-5, 1.25, -1.25])
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    src, tgt = rng.integers(0, N, E).astype(np.int32), rng.integers(0, N, E).astype(np.int32)
    ptr = _groups(rng, sizes)
    ref = torch.zeros(E, Dm, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        ref[lo:hi] = torch.cat([h[src[lo:hi]], h[tgt[lo:hi]]], -1).double() @ W[t].double()
    hp = ops.pack_f16x2(_dev(h), ops.H3_ROW_SCALE)
    recon = _unpack_f16x2(hp, Din) / ops.H3_ROW_SCALE
    tol = torch.maximum(h.double().abs() * 2.0 ** -23, torch.full_like(h.double(), 2.0 ** -25 / ops.H3_ROW_SCALE))
    assert ((recon - h.double()).abs() <= tol).all()
    out = ops.gemm_rows_h3([(hp, _dev(src), Din), (hp, _dev(tgt), Din)], ops.pack_weights_h3(_dev(W), True), E, Dm,
                           out_scale=1.0 / (ops.H3_ROW_SCALE * ops.H3_W_SCALE), group_ptr=_dev(ptr), G=T)
    err = (out.cpu().double() - ref).abs().max()
    exact = ops.gemm_rows([(_dev(h), _dev(src)), (_dev(h), _dev(tgt))], _dev(W), E, Dm, b_group_stride=2 * Din * Dm, ldb=Dm, group_ptr=_dev(ptr), G=T)
    err32 = (exact.cpu().double() - ref).abs().max()
    x6 = ops.gemm_rows_x6([(ops.pack_bf16x3(_dev(h)), _dev(src), Din), (ops.pack_bf16x3(_dev(h)), _dev(tgt), Din)],
                          ops.pack_weights_x6(_dev(W), True), E, Dm, group_ptr=_dev(ptr), G=T)
    err6 = (x6.cpu().double() - ref).abs().max()
    print(f"f16x3 {float(err):.2e}  bf16x6 {float(err6):.2e}  exact fp32 MFMA {float(err32):.2e}")
    assert err < 2e-5 and err < 4 * err32 + 1e-6, (float(err), float(err32))
    # saturation instead of inf, NaN propagation -- and the events are counted on the device (the trainer reads the counter per epoch)
    hp = ops.pack_f16x2(_dev(h), ops.H3_ROW_SCALE)
    ops.h3_saturation_events(reset=True)
    ops.pack_f16x2(_dev(h), ops.H3_ROW_SCALE), ops.pack_weights_h3(_dev(W), True)
    assert ops.h3_saturation_events() == 0  # the operands above fit
    wild = torch.tensor([[1e30, -1e30, float("nan"), 3.0e2] + [0.0] * (Din - 4)])
    wp = _unpack_f16x2(ops.pack_f16x2(_dev(wild), ops.H3_ROW_SCALE), Din)[0]
    assert wp[0] == 65504.0 and wp[1] == -65504.0 and torch.isnan(wp[2]) and wp[3] == 65504.0  # (300 x 256 > 65504: saturated)
    assert ops.h3_saturation_events(reset=True) == 1 and ops.h3_saturation_events() == 0  # one packing thread (8 values) clamped
    ops.pack_weights_h3(_dev(torch.full((1, 32, 128), 2000.0)), True)  # weights beyond +-1023
    assert ops.h3_saturation_events(reset=True) == 32 * 128 // 8


@pytest.mark.parametrize("Din,Dm,sizes", [(32, 64, [130, 0, 1, 700, 64]), (128, 128, [2100, 5, 300]), (64, 256, [129, 128, 1500]),
                                           (256, 256, [700, 33, 0, 1]), (128, 96, [31, 2000])])
def test_routed_gemms_f16x3_match_fp64(ops, Din, Dm, sizes):
    """f16x3 input-gradient and weight-gradient GEMMs of the max-aggregated messages: the gradient operand is packed with a scale
    derived ON THE DEVICE from its amax (its magnitude -- here ~1e-6 -- is not known in advance), the consumers divide it out."""
    rng = np.random.default_rng(1)
    N, T, E = 97, len(sizes), int(sum(sizes))
    h = torch.tanh(torch.randn(N, Din)) * 1.25
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    gq = torch.randn(N, Dm) * torch.exp(torch.randn(N, 1) * 2.0) * 1e-6  # tiny, heavy-tailed over the rows
    src, tgt = rng.integers(0, N, E).astype(np.int32), rng.integers(0, N, E).astype(np.int32)
    ptr = _groups(rng, sizes)
    arg = np.full((N, Dm), -1, dtype=np.int32)
    for n in range(N):
        inc = np.nonzero(tgt == n)[0]
        if len(inc):
            arg[n] = rng.choice(inc, Dm)
    e_ids = torch.arange(E)[:, None]
    Gm = torch.where(torch.from_numpy(arg)[tgt.astype(np.int64)].long() == e_ids, gq[tgt.astype(np.int64)].double(), torch.zeros(E, Dm, dtype=torch.float64))
    A = torch.cat([h[src.astype(np.int64)], h[tgt.astype(np.int64)]], -1).double()
    ref_dW = torch.zeros(T, 2 * Din, Dm, dtype=torch.float64)
    ref_dA = torch.zeros(E, 2 * Din, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        ref_dW[t] = A[lo:hi].T @ Gm[lo:hi]
        ref_dA[lo:hi] = Gm[lo:hi] @ W[t].double().T
    d_gq = _dev(gq)
    am = ops.amax(d_gq)
    assert float(am) == float(gq.abs().max())
    hp, gqp = ops.pack_f16x2(_dev(h), ops.H3_ROW_SCALE), ops.pack_f16x2(d_gq, 1.0, amax=am)
    # the device-derived scale put the largest entry into [2^13, 2^14]
    top = float(_unpack_f16x2(gqp, Dm).abs().max())
    assert 2.0 ** 13 <= top <= 2.0 ** 14
    d_src, d_tgt, d_ptr = _dev(src), _dev(tgt), _dev(ptr)
    won = (arg[tgt] == np.arange(E)[:, None])
    bits = np.packbits(won.reshape(E, Dm // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(E, Dm // 32)
    d_bits = _dev(bits.view(np.int32))
    gw = torch.zeros(T, 2 * Din, Dm, device="cuda")
    ops.gemm_wgrad_h3([(hp, d_src, Din), (hp, d_tgt, Din)], gqp, E, Dm, gw, out_scale=1.0 / ops.H3_ROW_SCALE, g_idx=d_tgt, win_bits=d_bits,
                      g_amax=am, gw_group_stride=2 * Din * Dm, group_ptr=d_ptr, G=T)
    gw6 = torch.zeros_like(gw)
    ops.gemm_wgrad_routed_x6([(ops.pack_bf16x3(_dev(h)), d_src, Din), (ops.pack_bf16x3(_dev(h)), d_tgt, Din)], ops.pack_bf16x3(d_gq), d_tgt, d_bits,
                             E, Dm, gw6, gw_group_stride=2 * Din * Dm, group_ptr=d_ptr, G=T)
    scale = float(ref_dW.abs().max())
    err, err6 = float((gw.cpu().double() - ref_dW).abs().max()), float((gw6.cpu().double() - ref_dW).abs().max())
    print(f"wgrad: f16x3 {err / scale:.2e}  bf16x6 {err6 / scale:.2e} (relative to the largest entry {scale:.2e})")
    assert err < 4e-6 * scale, (err, err6, scale)
    dA = ops.gemm_rows_h3([(gqp, d_tgt, Dm)], ops.pack_weights_h3(_dev(W), False), E, 2 * Din, out_scale=1.0 / ops.H3_W_SCALE, a_amax=am,
                          group_ptr=d_ptr, G=T, win_bits=d_bits)
    sA = float(ref_dA.abs().max())
    assert float((dA.cpu().double() - ref_dA).abs().max()) < 4e-6 * sA, (float((dA.cpu().double() - ref_dA).abs().max()), sA)
    # unrouted weight gradient (plain rows): the dense form
    g2 = torch.randn(E, Dm) * 1e-3
    am2 = ops.amax(_dev(g2))
    gw2 = torch.zeros(T, 2 * Din, Dm, device="cuda")
    ops.gemm_wgrad_h3([(hp, d_src, Din), (hp, d_tgt, Din)], ops.pack_f16x2(_dev(g2), 1.0, amax=am2), E, Dm, gw2, out_scale=1.0 / ops.H3_ROW_SCALE,
                      g_amax=am2, gw_group_stride=2 * Din * Dm, group_ptr=d_ptr, G=T)
    ref2 = torch.stack([A[ptr[t]:ptr[t + 1]].T @ g2[ptr[t]:ptr[t + 1]].double() for t in range(T)])
    assert float((gw2.cpu().double() - ref2).abs().max()) < 4e-6 * float(ref2.abs().max())


@pytest.mark.parametrize("magnitude", [0.0, 1e-30, 1e-12, 1.0, 1e12, 1e30])
def test_f16x3_gradient_operand_scale_follows_any_magnitude(ops, magnitude):
    """The gradient operand of the f16x3 GEMMs is packed with a scale derived on the device from its amax: results are equally
    accurate RELATIVE to the tensor's size whether it is 1e-30 or 1e30 large (fp16's own range is 6e-8 ... 65504), an all-zero
    gradient gives exact zeros, and nothing saturates."""
    rng = np.random.default_rng(5)
    N, E, Din, Dm, T = 64, 700, 64, 96, 2
    h = torch.tanh(torch.randn(N, Din)) * 1.25
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    g = torch.randn(E, Dm) * magnitude
    src, tgt = rng.integers(0, N, E).astype(np.int32), rng.integers(0, N, E).astype(np.int32)
    ptr = np.array([0, 300, E], dtype=np.int32)
    ops.h3_saturation_events(reset=True)
    am = ops.amax(_dev(g))
    gp = ops.pack_f16x2(_dev(g), 1.0, amax=am)
    hp = ops.pack_f16x2(_dev(h), ops.H3_ROW_SCALE)
    gw = torch.zeros(T, 2 * Din, Dm, device="cuda")
    ops.gemm_wgrad_h3([(hp, _dev(src), Din), (hp, _dev(tgt), Din)], gp, E, Dm, gw, out_scale=1.0 / ops.H3_ROW_SCALE, g_amax=am,
                      gw_group_stride=2 * Din * Dm, group_ptr=_dev(ptr), G=T)
    dA = ops.gemm_rows_h3([(gp, None, Dm)], ops.pack_weights_h3(_dev(W), False), E, 2 * Din, out_scale=1.0 / ops.H3_W_SCALE, a_amax=am,
                          group_ptr=_dev(ptr), G=T)
    A = torch.cat([h[src.astype(np.int64)], h[tgt.astype(np.int64)]], -1).double()
    ref_w = torch.stack([A[ptr[t]:ptr[t + 1]].T @ g[ptr[t]:ptr[t + 1]].double() for t in range(T)])
    ref_a = torch.cat([g[ptr[t]:ptr[t + 1]].double() @ W[t].double().T for t in range(T)])
    assert ops.h3_saturation_events(reset=True) == 0
    if magnitude == 0.0:
        assert float(am) == 0.0 and not gw.any() and not dA.any()
        return
    assert torch.isfinite(gw).all() and torch.isfinite(dA).all()
    assert float((gw.cpu().double() - ref_w).abs().max()) < 4e-6 * float(ref_w.abs().max())
    assert float((dA.cpu().double() - ref_a).abs().max()) < 4e-6 * float(ref_a.abs().max())


@pytest.mark.parametrize("M,K,N", [(1000, 128, 128), (777, 256, 64), (130, 64, 96), (2500, 256, 160)])
def test_dense_bf16x6_gemms_match_fp64(ops, M, K, N):
    """The dense node update on the bf16 matrix cores: bl_gemm_rows_x6_epi (bias + tanh + counter-hash dropout epilogue, the
    SAME dropout counter as the exact-fp32 row GEMM), the plain input-gradient form and bl_gemm_wgrad_x6 (no routing)."""
    torch.manual_seed(3)
    x, W, b, g = torch.randn(M, K), torch.randn(K, N) / math.sqrt(K), torch.randn(N) * 0.1, torch.randn(M, N)
    xp, gp = ops.pack_bf16x3(_dev(x)), ops.pack_bf16x3(_dev(g))
    wkn, wnk = ops.pack_weights_x6(_dev(W).unsqueeze(0), True), ops.pack_weights_x6(_dev(W).unsqueeze(0), False)
    drop = ops.Dropout(0.25, 7, 3)
    # forward with epilogue, against the exact-fp32 kernel with the same epilogue (same dropout mask) and fp64
    out = ops.gemm_rows_x6([(xp, None, K)], wkn, M, N, bias=_dev(b), act=ops.ACT_TANH, drop=drop)
    exact = ops.gemm_rows([(_dev(x), None)], _dev(W), M, N, bias=_dev(b), act=ops.ACT_TANH, drop=drop)
    assert torch.equal(out == 0, exact == 0)  # identical keep mask
    assert float((out - exact).abs().max()) < 5e-6
    nodrop = ops.gemm_rows_x6([(xp, None, K)], wkn, M, N, bias=_dev(b), act=ops.ACT_TANH)
    ref = torch.tanh(x.double() @ W.double() + b.double())
    assert float((nodrop.cpu().double() - ref).abs().max()) < 5e-6
    # input gradient g . W^T
    gin = ops.gemm_rows_x6([(gp, None, N)], wnk, M, K)
    assert float((gin.cpu().double() - g.double() @ W.double().t()).abs().max()) < 2e-5
    # weight gradient x^T . g (accumulates)
    gw = torch.ones(K, N, device="cuda")
    ops.gemm_wgrad_x6([(xp, None, K)], gp, M, N, gw)
    want = 1.0 + x.double().t() @ g.double()
    assert float((gw.cpu().double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


def test_segment_max_and_act_bwd_packed_outputs(ops):
    """What the fused layer hands to the bf16x6 dense GEMMs: bl_mp_layer's internal packed LayerNorm output / packed g_z are
    bl_pack_bf16x3 of the fp32 results -- checked through the layer call by comparing the two dense paths end to end."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module

    mb = to_device(collate_samples(make_samples(3, seed=9, num_nodes=257, num_messages=1300, num_edge_types=5, vocab_size=400), 5), "cuda")
    torch.manual_seed(1)
    module = build_gnn_mlp_module(64, 4, 5, vocabulary_size=400, dropout_rate=0.2).cuda().train()
    res = {}
    for x6 in (True, False):
        hip_ops.DENSE_X6 = x6
        try:
            module.zero_grad(set_to_none=True)
            loss = module(**mb, dropout_seed=5)
            loss.backward()
            hip_ops.join_side_stream()
            torch.cuda.synchronize()
            res[x6] = (float(loss.detach()), {k: p.grad.clone() for k, p in module.named_parameters()})
        finally:
            hip_ops.DENSE_X6 = True
    assert abs(res[True][0] - res[False][0]) < 2e-6
    for k, g in res[True][1].items():
        ref = res[False][1][k]
        assert float((g - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-7, k  # (routing near-ties may flip: see test_hip_parity)


@pytest.mark.parametrize("Din,Dm,sizes", [(32, 64, [130, 0, 1, 700, 64]), (128, 128, [2100, 5, 300]), (64, 256, [129, 128, 1500]),
                                           (256, 256, [700, 33, 0, 1]), (128, 96, [31, 2000])])
def test_routed_gemms_bf16x6_match_fp64(ops, Din, Dm, sizes):
    """bf16x6 input-gradient and weight-gradient GEMMs of the max-aggregated messages (winner-masked
    operand, transposing LDS reads in the weight gradient) against an fp64 reference."""
    rng = np.random.default_rng(1)
    N, T, E = 97, len(sizes), int(sum(sizes))
    h = torch.randn(N, Din)
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    gq = torch.randn(N, Dm)
    src, tgt = rng.integers(0, N, E).astype(np.int32), rng.integers(0, N, E).astype(np.int32)
    ptr = _groups(rng, sizes)
    # winner table: for every (node, channel) one of the messages targeting the node (or -1)
    arg = np.full((N, Dm), -1, dtype=np.int32)
    for n in range(N):
        inc = np.nonzero(tgt == n)[0]
        if len(inc):
            arg[n] = rng.choice(inc, Dm)
    Gm = torch.zeros(E, Dm, dtype=torch.float64)   # routed message gradient
    e_ids = torch.arange(E)[:, None]
    Gm = torch.where(torch.from_numpy(arg)[tgt.astype(np.int64)].long() == e_ids, gq[tgt.astype(np.int64)].double(), Gm)
    A = torch.cat([h[src.astype(np.int64)], h[tgt.astype(np.int64)]], -1).double()
    ref_dW = torch.zeros(T, 2 * Din, Dm, dtype=torch.float64)
    ref_dA = torch.zeros(E, 2 * Din, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        ref_dW[t] = A[lo:hi].T @ Gm[lo:hi]
        ref_dA[lo:hi] = Gm[lo:hi] @ W[t].double().T
    hp, gqp = ops.pack_bf16x3(_dev(h)), ops.pack_bf16x3(_dev(gq))
    d_arg, d_src, d_tgt, d_ptr = _dev(arg), _dev(src), _dev(tgt), _dev(ptr)
    won = (arg[tgt] == np.arange(E)[:, None])                       # [E, Dm] routing as booleans
    bits = np.packbits(won.reshape(E, Dm // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(E, Dm // 32)
    d_bits = _dev(bits.view(np.int32))
    gw = torch.zeros(T, 2 * Din, Dm, device="cuda")
    ops.gemm_wgrad_routed_x6([(hp, d_src, Din), (hp, d_tgt, Din)], gqp, d_tgt, d_bits, E, Dm, gw, gw_group_stride=2 * Din * Dm,
                             group_ptr=d_ptr, G=T)
    gw32 = torch.zeros_like(gw)
    ops.gemm_wgrad_routed([(_dev(h), d_src), (_dev(h), d_tgt)], _dev(gq), d_tgt, d_arg, E, Dm, gw32, gw_group_stride=2 * Din * Dm,
                          group_ptr=d_ptr, G=T)
    scale = float(ref_dW.abs().max())
    err, err32 = float((gw.cpu().double() - ref_dW).abs().max()), float((gw32.cpu().double() - ref_dW).abs().max())
    assert err < 2e-6 * max(scale, 1.0) * 4 and err < 4 * err32 + 1e-6 * scale, (err, err32, scale)
    # the 128 x 128 tile and the wide 256 x 128 tile (taken above when 2 Din is a multiple of 256 and Din of 128) do the same
    # products in the same order per tile: equal up to the order of the chunks' atomic adds
    prev = ops.set_wgrad_tile(128)
    try:
        gw128 = torch.zeros_like(gw)
        ops.gemm_wgrad_routed_x6([(hp, d_src, Din), (hp, d_tgt, Din)], gqp, d_tgt, d_bits, E, Dm, gw128, gw_group_stride=2 * Din * Dm,
                                 group_ptr=d_ptr, G=T)
    finally:
        ops.set_wgrad_tile(prev)
    assert prev == 256 and float((gw128 - gw).abs().max()) < 1e-5 * max(scale, 1.0)
    assert float((gw128.cpu().double() - ref_dW).abs().max()) < 2e-6 * max(scale, 1.0) * 4
    dA = ops.gemm_rows_x6([(gqp, d_tgt, Dm)], ops.pack_weights_x6(_dev(W), False), E, 2 * Din, group_ptr=d_ptr, G=T, win_bits=d_bits)
    assert float((dA.cpu().double() - ref_dA).abs().max()) < 2e-5


@pytest.mark.parametrize("widths,N,sizes", [((128, 128), 256, [130, 0, 1, 700, 64]), ((256, 256), 256, [2100, 5, 300]),
                                            ((32, 32), 512, [129, 128, 127]), ((64, 64, 128), 256, [513, 77]),
                                            ((256,), 512, [1000])])
def test_wide_row_gemm_is_bit_identical_to_the_128_tile(ops, widths, N, sizes):
    """bl_gemm_rows_x6w (128 x 256 tile, LDS-DMA operands, its own weight image) is a second schedule of bl_gemm_rows_x6's
    computation: same products, same summation order per element -> bit-identical results.  Plain form with 1..3 gathered
    sources, ragged / empty groups, row tails; routed form (one gathered source, winner-masked) on the same shapes."""
    rng = np.random.default_rng(5)
    K, T, E, R = int(sum(widths)), len(sizes), int(sum(sizes)), 301
    # the shape predicate of the layer calls: >= 256 output columns in multiples of 256, K >= 256 in multiples of 64 (the entry
    # point itself also takes shorter K: the (32, 32) case)
    assert ops.rows_x6w_ok(N, K) == (K >= 256) and not ops.rows_x6w_ok(N + 32, K) and not ops.rows_x6w_ok(N, K + 32)
    ptr = _dev(_groups(rng, sizes))
    W = torch.randn(T, K, N) / math.sqrt(K)
    srcs = []
    for w in widths:
        x = torch.randn(R, w)
        srcs.append((ops.pack_bf16x3(_dev(x)), _dev(rng.integers(0, R, E).astype(np.int32)), w))
    tiled, wide = ops.pack_weights_x6(_dev(W), True), ops.pack_weights_x6w(_dev(W), True)
    ref = ops.gemm_rows_x6(srcs, tiled, E, N, group_ptr=ptr, G=T)
    got = ops.gemm_rows_x6(srcs, wide, E, N, group_ptr=ptr, G=T, wide=True)
    assert torch.equal(got, ref)
    # against fp64 once more (the reference kernel has its own test; this guards the harness)
    A = torch.cat([torch.from_numpy(_unpack3(xp.cpu().numpy(), w))[idx.cpu().long()] for xp, idx, w in srcs], -1).double()
    want = torch.cat([A[int(ptr[t]):int(ptr[t + 1])] @ W[t].double() for t in range(T)])
    assert float((got.cpu().double() - want).abs().max()) < 5e-5
    # no groups at all: plain row tiles, ungathered single source
    if len(widths) == 1:
        xp = srcs[0][0]
        assert torch.equal(ops.gemm_rows_x6([(xp, None, K)], wide[:1], R, N, wide=True), ops.gemm_rows_x6([(xp, None, K)], tiled[:1], R, N))
    # routed form: C = (G_r masked by the winner bits) . W^T with W stored [T][N][K] (w_is_kn = 0)
    gq = torch.randn(R, K)
    bits = _dev(rng.integers(-2 ** 31, 2 ** 31, (E, K // 32)).astype(np.int32))
    bits[::7] = 0   # messages that won nothing
    bits[3::11] = -1  # ... or everything
    gqp, tgt = ops.pack_bf16x3(_dev(gq)), _dev(rng.integers(0, R, E).astype(np.int32))
    Wnk = torch.randn(T, N, K) / math.sqrt(K)
    r_ref = ops.gemm_rows_x6([(gqp, tgt, K)], ops.pack_weights_x6(_dev(Wnk), False), E, N, group_ptr=ptr, G=T, win_bits=bits)
    r_got = ops.gemm_rows_x6([(gqp, tgt, K)], ops.pack_weights_x6w(_dev(Wnk), False), E, N, group_ptr=ptr, G=T, win_bits=bits, wide=True)
    assert torch.equal(r_got, r_ref)
    prev = ops.load_library().bl_set_rows_tile(128)  # the measurement switch turns the shape predicate off
    try:
        assert prev == 256 and not ops.rows_x6w_ok(N, 512)
    finally:
        ops.load_library().bl_set_rows_tile(prev)


def _unpack3(packed: np.ndarray, width: int) -> np.ndarray:
    """bl_pack_bf16x3 rows [R, 3 * width] int16 -> fp32 [R, width] (hi + mid + lo)."""
    u = packed.view(np.uint16).reshape(packed.shape[0], 3, width).astype(np.uint32) << 16
    f = u.view(np.float32)
    return (f[:, 0].astype(np.float64) + f[:, 1] + f[:, 2]).astype(np.float32)


@pytest.mark.parametrize("Din,Dm,sizes,split", [(64, 64, [130, 0, 1, 700, 64], None), (128, 128, [2100, 5, 300], None),
                                                 (128, 128, [900, 0, 40], 64), (64, 128, [129, 1500], 32), (128, 64, [77, 300], None)])
def test_routed_input_gradient_from_the_nonzeros_matches_fp64(ops, Din, Dm, sizes, split):
    """bl_routed_dgrad_vec (per-message rows), bl_routed_dgrad_nodes (all-atomic node sums) and bl_routed_dgrad_nodes_rows
    (target halves by atomics, source halves as rows + bl_mp_scatter_grad over the source CSR) against an fp64 reference of
    the autograd of gather + per-type Linear + scatter_max; messages target-sorted inside a type with runs of equal targets,
    empty types, messages that won nothing, and the two-output (folded ConcatResidual) form."""
    rng = np.random.default_rng(7)
    N, T, E = 61, len(sizes), int(sum(sizes))
    assert ops.load_library().bl_routed_dgrad_vec_ok(Dm, 2 * Din) == 1
    W = torch.randn(T, 2 * Din, Dm) / math.sqrt(2 * Din)
    gq = torch.randn(N, Dm)
    ptr = _groups(rng, sizes)
    tgt = np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes] + [np.zeros(0, np.int64)]).astype(np.int32)
    src = rng.integers(0, N, E).astype(np.int32)
    arg = np.full((N, Dm), -1, dtype=np.int32)
    for n in range(N):
        inc = np.nonzero(tgt == n)[0]
        if len(inc):
            arg[n] = rng.choice(inc[: max(1, len(inc) - 1)], Dm)  # (the last incoming message of a node wins nothing)
    won = (arg[tgt] == np.arange(E)[:, None])
    Gm = torch.where(torch.from_numpy(won), gq[tgt.astype(np.int64)].double(), torch.zeros(E, Dm, dtype=torch.float64))
    ref_dA = torch.zeros(E, 2 * Din, dtype=torch.float64)
    for t in range(T):
        lo, hi = ptr[t], ptr[t + 1]
        ref_dA[lo:hi] = Gm[lo:hi] @ W[t].double().T
    ref_h = torch.zeros(N, Din, dtype=torch.float64)
    ref_h.index_add_(0, torch.from_numpy(src.astype(np.int64)), ref_dA[:, :Din])
    ref_h.index_add_(0, torch.from_numpy(tgt.astype(np.int64)), ref_dA[:, Din:])
    bits = np.packbits(won.reshape(E, Dm // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(E, Dm // 32)
    d_bits, d_src, d_tgt, d_ptr, d_gq = _dev(bits.view(np.int32)), _dev(src), _dev(tgt), _dev(ptr), _dev(gq)
    wt = _dev(W).transpose(1, 2).contiguous()
    tol = 2e-6 * max(1.0, float(ref_h.abs().max()))

    dA = ops.routed_dgrad_vec(d_gq, d_tgt, d_bits, d_ptr, T, wt, E, 2 * Din)
    assert float((dA.cpu().double() - ref_dA).abs().max()) < 2e-6 * max(1.0, float(ref_dA.abs().max()))
    dA2 = ops.routed_dgrad_vec(d_gq, d_tgt, d_bits, d_ptr, T, wt, E, 2 * Din)
    assert torch.equal(dA, dA2)  # bit-reproducible: the deterministic mode relies on it

    def outputs():
        if split is None:
            return torch.zeros(N, Din, device="cuda"), None
        return torch.zeros(N, split, device="cuda"), torch.zeros(N, Din - split, device="cuda")

    def joined(lo, hi):
        return (lo if hi is None else torch.cat([lo, hi], 1)).cpu().double()

    lo, hi = outputs()
    ops.routed_dgrad_nodes(d_gq, d_src, d_tgt, d_bits, d_ptr, T, wt, E, Din, lo, hi)
    assert float((joined(lo, hi) - ref_h).abs().max()) < tol

    lo, hi = outputs()
    rows = torch.full((E, Din), float("nan"), device="cuda")  # every row must be written, also for messages that won nothing
    ops.routed_dgrad_nodes(d_gq, d_src, d_tgt, d_bits, d_ptr, T, wt, E, Din, lo, hi, src_rows=rows)
    assert float((rows.cpu().double() - ref_dA[:, :Din]).abs().max()) < 2e-6 * max(1.0, float(ref_dA.abs().max()))
    ref_tgt = torch.zeros(N, Din, dtype=torch.float64).index_add_(0, torch.from_numpy(tgt.astype(np.int64)), ref_dA[:, Din:])
    assert float((joined(lo, hi) - ref_tgt).abs().max()) < tol  # the target halves alone so far
    if split is None:  # the second step as the fused layer issues it (the split form is covered by the layer tests)
        order = np.argsort(src, kind="stable").astype(np.int32)
        sptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=N))]).astype(np.int32)
        lib = ops.load_library()
        d_sptr, d_order = _dev(sptr), _dev(order)  # (named: a temporary's memory may be handed to the next allocation before the launch)
        ops._check(lib.bl_mp_scatter_grad(rows.data_ptr(), rows.stride(0), d_sptr.data_ptr(), d_order.data_ptr(), None, None, N, Din,
                                          1, lo.data_ptr(), lo.stride(0), None, ops._stream()), "bl_mp_scatter_grad")
        assert float((lo.cpu().double() - ref_h).abs().max()) < tol


def test_gather_rows_fwd_bwd(ops):
    rng = np.random.default_rng(3)
    x = torch.randn(300, 64)
    idx = rng.integers(0, 300, 777).astype(np.int32)
    xd = _dev(x).requires_grad_(True)
    out = ops.gather_rows(xd, _dev(idx))
    assert torch.equal(out.cpu(), x[idx.astype(np.int64)])
    g = torch.randn(777, 64)
    out.backward(_dev(g))
    ref = torch.zeros(300, 64).index_add_(0, torch.from_numpy(idx.astype(np.int64)), g)
    assert (xd.grad.cpu() - ref).abs().max() < 1e-5


@pytest.mark.parametrize("placement", ["aggregated", "message"])
def test_fused_layer_call_equals_kernel_by_kernel_path(placement):
    """bl_mp_layer_fwd / bl_mp_layer_bwd (one C call per layer and direction, ConcatResidual input read as a
    pair) against the same kernels driven one by one from Python with a materialised concatenation."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module

    mb = to_device(collate_samples(make_samples(3, seed=5, num_nodes=300, num_messages=1500, num_edge_types=6, vocab_size=500), 6), "cuda")
    torch.manual_seed(0)
    module = build_gnn_mlp_module(64, 8, 6, vocabulary_size=500, dropout_rate=0.1, message_activation_placement=placement).cuda().train()
    res = {}
    # (the kernel-by-kernel path runs its message GEMMs as bf16x6: the fused call is put on the same split, so that "same kernels,
    # same order" holds bit for bit; the default f16x3 split of the fused call is compared with it at the end)
    prev_split = hip_ops.set_msg_gemm_mode("bf16x6")
    for fused in (True, False, "f16x3"):
        hip_ops.FUSED_LAYER = bool(fused)
        hip_ops.DENSE_X6 = False  # the kernel-by-kernel path runs the dense node update on the exact-fp32 GEMMs
        if fused == "f16x3":
            hip_ops.set_msg_gemm_mode("f16x3")
        try:
            module.zero_grad(set_to_none=True)
            loss = module(**mb, dropout_seed=11)
            loss.backward()
            hip_ops.join_side_stream()
            torch.cuda.synchronize()
            res[fused] = (float(loss.detach()), {k: p.grad.clone() for k, p in module.named_parameters()})
        finally:
            hip_ops.FUSED_LAYER = True
            hip_ops.DENSE_X6 = True
            if fused == "f16x3":
                hip_ops.set_msg_gemm_mode(prev_split)
    assert res[True][0] == res[False][0]  # same kernels, same order: bit-identical forward
    for k, g in res[True][1].items():
        ref = res[False][1][k]
        assert float((g - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-7, k  # atomics reorder sums
    # the f16x3 split: another fp32-accurate evaluation of the same products (a near-tie of the routed max may flip: 1e-3)
    assert abs(res["f16x3"][0] - res[False][0]) < 1e-5
    for k, g in res["f16x3"][1].items():
        ref = res[False][1][k]
        assert float((g - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-7, k


def _unpack_bf16x3(p, D):
    """fp64 value of a bl_pack_bf16x3 row-packed int16 tensor [R, 3 D]: hi + mid + lo."""
    u = p.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    f = torch.from_numpy(u.view(np.float32).astype(np.float64)).view(p.shape[0], 3, D)
    return f.sum(1)


@pytest.mark.parametrize("N,Dm,Dout,p,with_dact,want_f32", [(1000, 128, 128, 0.2, True, True), (777, 256, 128, 0.0, True, False),
                                                             (300, 256, 256, 0.1, False, True), (64, 128, 256, 0.2, True, True),
                                                             (65, 128, 64, 0.0, True, True), (1, 256, 32, 0.3, True, True)])
def test_node_update_bwd_matches_fp64(ops, N, Dm, Dout, p, with_dact, want_f32):
    """bl_node_update_bwd (act backward -> dense input gradient -> LayerNorm backward x activation derivative, one kernel) against
    the same chain in fp64: packed g_z, the bias gradient, gq in both forms, the gamma / beta gradients (accumulated on top of
    what is there).  Partial last tile, one row, Dout below one 128-wide tile, no activation derivative."""
    torch.manual_seed(N + Dm)
    drop = ops.Dropout(p, 123, 5) if p > 0 else ops.NO_DROPOUT
    keep = torch.from_numpy(O.dropout_keep_mask(123, 5, N * Dout, p)).view(N, Dout).double() if p > 0 else torch.ones(N, Dout, dtype=torch.float64)
    t = torch.tanh(torch.randn(N, Dout).double())
    y = (t * keep / (1 - p)).float()
    g = torch.randn(N, Dout)
    Wd = torch.randn(Dm, Dout) / math.sqrt(Dm)  # forward weight [Dm, Dout]: z = ln_out @ Wd
    agg = torch.randn(N, Dm) * 1.5 + 0.3
    gamma = torch.rand(Dm) + 0.5
    dact = torch.rand(N, Dm) if with_dact else None
    mean = agg.double().mean(1)
    var = agg.double().var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    # fp64 chain
    gz = g.double() * keep / (1 - p) * (1 - t * t)
    gln = gz @ Wd.double().t()
    xh = (agg.double() - mean[:, None]) * rstd[:, None]
    gg = gln * gamma.double()
    a, b = gg.mean(1, keepdim=True), (gg * xh).mean(1, keepdim=True)
    gq_ref = rstd[:, None] * (gg - a - xh * b) * (dact.double() if with_dact else 1.0)
    # device
    wnk = ops.pack_weights_x6(_dev(Wd).unsqueeze(0), False)
    g_bias = torch.full((Dout,), 2.0, device="cuda")
    g_lng, g_lnb = torch.full((Dm,), -1.0, device="cuda"), torch.full((Dm,), 0.5, device="cuda")
    assert ops.load_library().bl_node_update_bwd_ok(Dm, Dout)
    gzp, gq, gqp = ops.node_update_bwd(_dev(g), _dev(y), drop, wnk, _dev(agg), _dev(mean.float()), _dev(rstd.float()), _dev(gamma),
                                       _dev(dact) if with_dact else None, g_bias, g_lng, g_lnb, want_f32=want_f32)
    torch.cuda.synchronize()
    scale = float(gq_ref.abs().max())
    assert float((_unpack_bf16x3(gzp, Dout) - gz).abs().max()) < 1e-6 * max(1.0, float(gz.abs().max()))
    assert float((g_bias.cpu().double() - (2.0 + gz.sum(0))).abs().max()) < 1e-4 * max(1.0, float(gz.sum(0).abs().max()))
    if want_f32:
        assert float((gq.cpu().double() - gq_ref).abs().max()) < 2e-5 * max(1.0, scale)
    else:
        assert gq is None
    assert float((_unpack_bf16x3(gqp, Dm) - gq_ref).abs().max()) < 2e-5 * max(1.0, scale)
    want_g, want_b = -1.0 + (gln * xh).sum(0), 0.5 + gln.sum(0)
    assert float((g_lng.cpu().double() - want_g).abs().max()) < 1e-4 * max(1.0, float(want_g.abs().max()))
    assert float((g_lnb.cpu().double() - want_b).abs().max()) < 1e-4 * max(1.0, float(want_b.abs().max()))


@pytest.mark.parametrize("hidden", [128, 64])
def test_fused_node_update_backward_equals_the_three_kernel_chain(hidden):
    """bl_mp_layer_bwd with bl_node_update_bwd (default) against the same call with the three kernels it replaces
    (bl_set_fused_node_bwd(0)): same loss, gradients equal up to the order of fp32 sums.  hidden 128: layers with Dm = 128 and
    the ConcatResidual layers' Dm = 256 both take the fused kernel; hidden 64: Dm = 64 does not qualify, Dm = 128 does."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module

    mb = to_device(collate_samples(make_samples(3, seed=5, num_nodes=300, num_messages=1500, num_edge_types=6, vocab_size=500), 6), "cuda")
    torch.manual_seed(0)
    module = build_gnn_mlp_module(hidden, 8, 6, vocabulary_size=500, dropout_rate=0.1).cuda().train()
    res = {}
    for fused in (True, False):
        prev = hip_ops.set_fused_node_bwd(fused)
        try:
            module.zero_grad(set_to_none=True)
            loss = module(**mb, dropout_seed=11)
            loss.backward()
            hip_ops.join_side_stream()
            torch.cuda.synchronize()
            res[fused] = (float(loss.detach()), {k: p.grad.clone() for k, p in module.named_parameters()})
        finally:
            hip_ops.set_fused_node_bwd(prev)
    assert res[True][0] == res[False][0]
    for k, g in res[True][1].items():
        ref = res[False][1][k]
        assert float((g - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7, k


@pytest.mark.parametrize("placement", ["aggregated", "message"])
@pytest.mark.parametrize("degree", ["uniform", "powerlaw"])
def test_forward_only_layer_call_equals_the_training_forward(degree, placement):
    """bl_mp_layer_fwd with saved == NULL (what model.predict / evaluate.py run under no_grad: reference
    buglab/models/gnn.py:606-645) keeps nothing for a backward pass; its outputs are those of the training-form call, bit
    for bit -- hubs (4-wave segmented max) and a ConcatResidual input included."""
    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module

    mb = to_device(collate_samples(make_samples(3, seed=9, num_nodes=400, num_messages=2400, num_edge_types=6, vocab_size=500,
                                                degree=degree, max_degree=200), 6), "cuda")
    torch.manual_seed(1)
    module = build_gnn_mlp_module(64, 8, 6, vocabulary_size=500, dropout_rate=0.1, message_activation_placement=placement).cuda().eval()
    res = {}
    for infer in (True, False):
        hip_ops.INFERENCE_MODE = infer
        try:
            with torch.no_grad():
                ids, logp, gout, _ = module.compute_localization_logprobs(mb["graph_data"])
            torch.cuda.synchronize()
            res[infer] = (logp.clone(), gout.output_node_representations.clone())
        finally:
            hip_ops.INFERENCE_MODE = True
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert torch.isfinite(res[True][0]).all()


def test_forward_only_layer_call_without_edges():
    """saved == NULL and E == 0: the workspace still holds the packed input and the LayerNorm output."""
    from buglab.models import hip_ops

    N, H, T = 37, 64, 3
    torch.manual_seed(0)
    h = torch.randn(N, H, device="cuda")
    W = torch.randn(T, 2 * H, H, device="cuda") * 0.1
    ln_g, ln_b = torch.rand(H, device="cuda") + 0.5, torch.randn(H, device="cuda") * 0.1
    Wd, bd = torch.randn(H, H, device="cuda") * 0.1, torch.randn(H, device="cuda") * 0.1
    z = torch.zeros(0, dtype=torch.int32, device="cuda")
    zp = torch.zeros(N + 1, dtype=torch.int32, device="cuda")
    g = hip_ops.GraphIndex(z, z, torch.zeros(T + 1, dtype=torch.int32, device="cuda"), zp, z, zp, z, N, 0, T)
    outs = []
    for infer in (True, False):
        hip_ops.INFERENCE_MODE = infer
        try:
            with torch.no_grad():
                outs.append(hip_ops.mp_layer(h, W, ln_g, ln_b, Wd, bd, g, msg_act="gelu"))
        finally:
            hip_ops.INFERENCE_MODE = True
    torch.cuda.synchronize()
    # every aggregate is 0: LayerNorm of a zero row is its bias
    ref = torch.tanh(ln_b[None, :].expand(N, H) @ Wd + bd)
    assert torch.equal(outs[0], outs[1]) and float((outs[0] - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("w_buggy,abstain,ncand", [(1.0, 0.0, 9), (2.5, 0.0, 9), (1.0, 0.35, 9), (0.4, 0.2, 9), (2.5, 0.35, 90)])
def test_fused_loss_assembly_equals_the_op_by_op_path(w_buggy, abstain, ncand):
    """hip_ops.bug_loss (one kernel per direction for localizationmodule.py:63-124 + gnn.py:221-251,295-311 + the fixers'
    forward()s) against the same arithmetic done op by op (which the reference-generated goldens pin): loss, every
    parameter gradient, and every metric -- for a buggy-sample weight != 1 and an abstain weight too."""
    from functools import partial

    from buglab.data.collate import collate_samples, to_device
    from buglab.data.synthetic import make_samples
    from buglab.models import hip_ops
    from buglab.models.gnn import build_gnn_mlp_module, const_weight_schedule

    # (ncand = 90: location groups of more than 64 entries take the loss kernels' looping path, the others their one-item-per-lane path)
    mb = to_device(collate_samples(make_samples(7, seed=21, num_nodes=max(90, 2 * ncand), num_messages=400, num_edge_types=5, vocab_size=300,
                                                num_candidates=ncand), 5), "cuda")
    res = {}
    for fused in (True, False):
        torch.manual_seed(4)
        module = build_gnn_mlp_module(64, 4, 5, vocabulary_size=300, dropout_rate=0.0, buggy_samples_weight=w_buggy).cuda().train()
        module._localization_module._abstain_weight = abstain
        hip_ops.FUSED_LOSS = fused
        try:
            losses = []
            for _ in range(2):  # two steps: the counters accumulate
                module.zero_grad(set_to_none=True)
                loss = module(**mb, dropout_seed=3)
                loss.backward()
                losses.append(float(loss.detach()))
            hip_ops.join_side_stream()
            torch.cuda.synchronize()
            res[fused] = (losses, {k: p.grad.clone() for k, p in module.named_parameters()}, module.report_metrics())
            module.reset_metrics()
            assert module.report_metrics() == {}
        finally:
            hip_ops.FUSED_LOSS = True
    assert max(abs(a - b) for a, b in zip(res[True][0], res[False][0])) < 2e-6
    for k, g in res[True][1].items():
        ref = res[False][1][k]
        assert float((g - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-8, k
    m1, m0 = res[True][2], res[False][2]
    assert set(m1) == set(m0)
    for k in m0:
        if isinstance(m0[k], str):
            assert m1[k] == m0[k], k
        else:
            assert abs(m1[k] - m0[k]) < 1e-5 or (m1[k] != m1[k] and m0[k] != m0[k]), (k, m1[k], m0[k])

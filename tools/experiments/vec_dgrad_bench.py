"""Routed input gradient: the shipped matrix-core GEMM vs the two vector-unit forms (non-zeros only), c2 H=128 layer shape.
Build first: tools/experiments/build_vec.sh.  Checks both forms against the shipped kernel, then times all three."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np, torch
from buglab.models import hip_ops as ops

V, I32 = ctypes.c_void_p, ctypes.c_int32
SIG = [V, I32, V, V, I32, V, I32, V, I32, I32, I32, V, I32, V]
libs = {}
for name, sym in (("vec1", "bl_routed_dgrad_vec"), ("vec2", "bl_routed_dgrad_vec2")):
    path = os.path.join(HERE, f"lib{name}.so")
    if os.path.exists(path):
        fn = getattr(ctypes.CDLL(path), sym)
        fn.argtypes, fn.restype = SIG, ctypes.c_int
        libs[name] = fn


def run(fn, gq, tgt, bits, ptr, T, wt, E, K2):
    out = torch.empty(E, K2, device="cuda")
    rc = fn(gq.data_ptr(), gq.stride(0), tgt.data_ptr(), bits.data_ptr(), bits.stride(0), ptr.data_ptr(), T, wt.data_ptr(), E, gq.shape[1], K2,
            out.data_ptr(), K2, None)
    assert rc == 0, rc
    return out


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


def main():
    N, E, Din, Dm, T = 128000, 640000, 128, 128, 16
    rng = np.random.default_rng(0)
    w = 1.0 / np.arange(1, T + 1); sizes = np.floor(w / w.sum() * E).astype(np.int64); sizes[0] += E - sizes.sum()
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    tgt = torch.from_numpy(np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes]).astype(np.int32)).cuda()
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
    wt = W.transpose(1, 2).contiguous()
    gqp, wp = ops.pack_bf16x3(gq), ops.pack_weights_x6(W, False)
    ref = lambda: ops.gemm_rows_x6([(gqp, tgt, Dm)], wp, E, 2 * Din, group_ptr=ptr, G=T, win_bits=bits)
    want = ref()
    scale = float(want.abs().max())
    for name, fn in libs.items():
        got = run(fn, gq, tgt, bits, ptr, T, wt, E, 2 * Din)
        torch.cuda.synchronize()
        print(f"{name}: max |diff| vs the shipped routed GEMM {float((got - want).abs().max()):.3e} (largest entry {scale:.3f})")
    nnz = int(won.sum())
    print(f"non-zeros of the message gradient: {nnz} of {E * Dm} ({nnz / (E * Dm):.1%})")
    print(f"shipped routed bf16x6 GEMM: {timeit(ref):.3f} ms")
    for name, fn in libs.items():
        print(f"{name}: {timeit(lambda: run(fn, gq, tgt, bits, ptr, T, wt, E, 2 * Din)):.3f} ms")


if __name__ == "__main__":
    main()

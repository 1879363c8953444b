"""v4 (ping-pong) bf16x6 row GEMM vs the shipped kernel: bit-identity and time at the c2 layer shapes.
Build first:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I<csrc> -shared bl_gemm_x6v4.hip -o libx6v4.so"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np, torch
from buglab.models import hip_ops as ops

lib = ctypes.CDLL(os.path.join(HERE, os.environ.get("V4LIB", "libx6v4.so")))
V = ctypes.c_void_p
lib.bl_pack_weights_x6v4.argtypes = [V, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, V, V]
lib.bl_gemm_rows_x6v4.argtypes = [ctypes.POINTER(ops.bl_rows_packed_t), V, ctypes.c_int64, V, V, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                  ctypes.c_int32, V, ctypes.c_int32, ctypes.c_int32, V]


def pack_v4(W, w_is_kn):
    G, K, N = (W.shape[0], W.shape[1], W.shape[2]) if w_is_kn else (W.shape[0], W.shape[2], W.shape[1])
    out = torch.zeros(G, ((N + 127) // 128) * (K // 32) * 12288, dtype=torch.int16, device="cuda")
    assert lib.bl_pack_weights_x6v4(W.data_ptr(), G, K, N, int(w_is_kn), out.data_ptr(), None) == 0
    return out


def gemm_v4(sources, bp, M, N, ptr, G, pingpong=1):
    r = ops.bl_rows_packed_t()
    K = 0
    for j, (xp, idx, width) in enumerate(sources):
        r.xp[j], r.idx[j], r.width[j] = xp.data_ptr(), (idx.data_ptr() if idx is not None else None), width
        K += width
    r.nsrc = len(sources)
    out = torch.empty(M, N, device="cuda")
    rc = lib.bl_gemm_rows_x6v4(ctypes.byref(r), bp.data_ptr(), bp.stride(0), ptr.data_ptr(), None, G, M, N, K, out.data_ptr(), N, pingpong, None)
    assert rc == 0, rc
    return out


def timeit(f, iters=10):
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
    N, E, Din, Dm, T = int(os.environ.get("NNODES", 128000)), int(os.environ.get("NMSGS", 640000)), int(os.environ.get("DIN", 128)), int(os.environ.get("DM", 128)), 16
    rng = np.random.default_rng(0)
    w = 1.0 / np.arange(1, T + 1); sizes = np.floor(w / w.sum() * E).astype(np.int64); sizes[0] += E - sizes.sum()
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    tgt = np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes]).astype(np.int32)
    src = (tgt // 2000 * 2000 + rng.integers(0, min(2000, N), E)).clip(0, N - 1).astype(np.int32)
    src, tgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    h = torch.randn(N, Din, device="cuda"); W = torch.randn(T, 2 * Din, Dm, device="cuda") / 16
    hp, wp1, wp4 = ops.pack_bf16x3(h), ops.pack_weights_x6(W, True), pack_v4(W, True)
    srcs = [(hp, src, Din), (hp, tgt, Din)]
    ref = ops.gemm_rows_x6(srcs, wp1, E, Dm, group_ptr=ptr, G=T)
    flop = 2.0 * E * 2 * Din * Dm
    for pp in (1, 0):
        for rep in range(3):  # a race would not show every time
            got = gemm_v4(srcs, wp4, E, Dm, ptr, T, pp)
            torch.cuda.synchronize()
            same = torch.equal(got, ref)
            if not same:
                bad = (got != ref)
                rows = bad.any(1).nonzero().flatten()
                print(f"pingpong={pp} rep {rep}: MISMATCH in {int(bad.sum())} entries, {rows.numel()} rows, first rows {rows[:8].tolist()}, max |d| {float((got - ref).abs().max()):.3e}")
                break
        else:
            print(f"pingpong={pp}: bit-identical to bl_gemm_rows_x6 (3 runs)")
    print(f"shape: N={N} E={E} K={2*Din} Dm={Dm} lib={os.environ.get('V4LIB', 'libx6v4.so')}")
    t1 = timeit(lambda: ops.gemm_rows_x6(srcs, wp1, E, Dm, group_ptr=ptr, G=T))
    print(f"v1 forward  {t1:.3f} ms  {flop / t1 / 1e9:.1f} TF/s ({flop / t1 / 1e9 / 416.7:.2f} of x6 peak)")
    for pp in (1, 0, 3, 2):  # 3 / 2: the same schedules without the in-loop DMA (time only)
        t4 = timeit(lambda: gemm_v4(srcs, wp4, E, Dm, ptr, T, pp))
        print(f"v4 pingpong={pp} {t4:.3f} ms  {flop / t4 / 1e9:.1f} TF/s ({flop / t4 / 1e9 / 416.7:.2f} of x6 peak)")
    if os.environ.get("TRACE"):
        trace(srcs, wp4, E, Dm, ptr, T)


def trace(srcs, wp4, E, Dm, ptr, T):
    lib.bl_v4_set_trace.argtypes = [V]
    dbg = torch.zeros(64 * 8 * 8, dtype=torch.int64, device="cuda")
    for pp in (1, 3):
        dbg.zero_()
        lib.bl_v4_set_trace(dbg.data_ptr())
        gemm_v4(srcs, wp4, E, Dm, ptr, T, pp)
        torch.cuda.synchronize()
        lib.bl_v4_set_trace(None)
        d = dbg.cpu().numpy().reshape(64, 8, 8)
        print(f"pingpong={pp}: 100 MHz ticks (10 ns) per workgroup: [locate+index loads, stage-0 DMA, k-loop, epilogue stores], waves 0 and 4")
        for b in range(0, 26, 5):
            for wv in (0, 4):
                x = d[b, wv]
                if x[0]:
                    print(f"  wg {b * 97 + 5:5d} wave {wv}: " + " ".join(f"{int(x[i + 1] - x[i]):6d}" for i in range(4)) + f"   total {int(x[4] - x[0])}")


if __name__ == "__main__":
    main()

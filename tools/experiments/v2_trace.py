"""Phase timeline of one workgroup of the v2 GEMM (s_memtime stamps), c2 forward shapes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np, torch
from buglab.models import hip_ops as ops
N, E, Din, Dm, T = 128000, 640000, 128, 128, 16
rng = np.random.default_rng(0)
w = 1.0 / np.arange(1, T + 1); sizes = np.floor(w / w.sum() * E).astype(np.int64); sizes[0] += E - sizes.sum()
ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
tgt = np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes]).astype(np.int32)
src = (tgt // 2000 * 2000 + rng.integers(0, 2000, E)).clip(0, N - 1).astype(np.int32)
src, tgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
h = torch.randn(N, Din, device="cuda"); W = torch.randn(T, 2 * Din, Dm, device="cuda") / 16
hp, wtp2 = ops.pack_bf16x3(h), ops.pack_weights_x6v2(W, True)
lib = ops.load_library()
lib.bl_v2_set_trace.argtypes = [ctypes.c_void_p]
dbg = torch.zeros(8 * 8 * 8, dtype=torch.int64, device="cuda")
tr = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for _ in range(3):
    ops.gemm_rows_x6v2([(hp, src, Din), (hp, tgt, Din)], wtp2, E, Dm, group_ptr=ptr, G=T, tile_rows=tr)
lib.bl_v2_set_trace(dbg.data_ptr())
ops.gemm_rows_x6v2([(hp, src, Din), (hp, tgt, Din)], wtp2, E, Dm, group_ptr=ptr, G=T, tile_rows=tr)
torch.cuda.synchronize()
lib.bl_v2_set_trace(None)
d = dbg.cpu().numpy().reshape(8, 8, 8)
t0 = d[:, 0, 0].min()
print("cycles (s_memtime, 100 MHz?) relative to first stamp; columns: top, reads done, dma issued, mfma issued, dma landed, after barrier")
for wv in range(tr // 32):
    for kt in range(8):
        print(f"wave {wv} stage {kt}:", " ".join(f"{int(x - t0):7d}" for x in d[wv, kt, :6]))

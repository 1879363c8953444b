"""GPU box: what clock / power does the chip sustain while one GEMM form runs back to back?  (rocm-smi sampled from a thread.)"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np, torch
from buglab.models import hip_ops as ops

N, E, Din, Dm, T = 64000, 320000, int(os.environ.get("DIN", 512)), int(os.environ.get("DM", 512)), 16
rng = np.random.default_rng(0)
w = 1.0 / np.arange(1, T + 1); sizes = np.floor(w / w.sum() * E).astype(np.int64); sizes[0] += E - sizes.sum()
ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
tgt = np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes]).astype(np.int32)
src = (tgt // 2000 * 2000 + rng.integers(0, 2000, E)).clip(0, N - 1).astype(np.int32)
src, tgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
h = torch.randn(N, Din, device="cuda"); W = torch.randn(T, 2 * Din, Dm, device="cuda") / 16
hp = ops.pack_bf16x3(h)
forms = {"x6 (128x128)": (ops.pack_weights_x6(W, True), False)}
if ops.rows_x6w_ok(Dm, 2 * Din):
    forms["x6w (128x256 dma)"] = (ops.pack_weights_x6w(W, True), True)
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.2)
for name, (wp, wide) in forms.items():
    f = lambda: ops.gemm_rows_x6([(hp, src, Din), (hp, tgt, Din)], wp, E, Dm, group_ptr=ptr, G=T, wide=wide)
    for _ in range(3): f()
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50): f()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    ms = e0.elapsed_time(e1) / n
    print(f"== {name}: K={2*Din} N={Dm}: {ms:.3f} ms per launch over {n} launches, {2.0*E*2*Din*Dm/ms/1e9:.1f} TF/s")
    import json, re
    for t, out in samples[2:10]:
        try:
            j = json.loads(out)
            c = j.get("card0", {})
            print("   ", {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower()})
        except Exception:
            print("   raw:", out[:300].replace("\n", " | "))

#!/usr/bin/env python
"""Per-stage timeline of ONE workgroup of the wide weight-gradient kernel (gemm_wgrad_x6_wide_kernel): builds a copy of the
library with -DBL_TRACE_WGRAD (shader-clock stamps at six points of every loop iteration, lane 0 of each wave), runs the
c2 layer shape and prints, per wave, the median cycles of each segment.   Run on the GPU box:  python tools/experiments/wgrad_trace.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd")
sys.path.insert(0, PKG)
import numpy as np
import torch

CSRC = os.path.join(PKG, "csrc")
ABLATE = int(os.environ.get("ABLATE", "0"))  # 1 no staging in the loop, 2 no MFMAs, 4 fragments read once (sums allowed); results are then wrong, times are not
TRACE = os.environ.get("TRACE", "1") != "0"
OUT = f"/tmp/bl_trace_{ABLATE}_{int(TRACE)}"
os.makedirs(OUT, exist_ok=True)
srcs = [f for f in os.listdir(CSRC) if f.endswith(".hip")]
objs = []
for f in srcs:
    o = os.path.join(OUT, f[:-4] + ".o")
    objs.append(o)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", f"-DBL_WW_ABLATE={ABLATE}"] + (["-DBL_TRACE_WGRAD"] if TRACE else []) + [
                    "-I", CSRC, "-I", os.path.join(CSRC, "..", "..", "include"),
                    "-c", (os.path.join(os.path.dirname(os.path.abspath(__file__)), "bl_gemm_x6_switches.hip") if f == "bl_gemm_x6.hip"
                           else os.path.join(CSRC, f)), "-o", o], check=True)  # (the switches live in the experiment twin of bl_gemm_x6.hip)
lib_path = os.path.join(OUT, "libbuglab_hip_trace.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path], check=True)

from buglab.models import hip_ops as ops

lib = ops.load_library(lib_path)
ops._lib = lib
if TRACE:
    lib.bl_debug_wgrad_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.bl_debug_wgrad_trace.restype = ctypes.c_int

rng = np.random.default_rng(0)
N, E, Din, Dm, T = 128000, 640000, int(os.environ.get("DIN", 128)), int(os.environ.get("DM", 128)), 16
w = 1.0 / np.arange(1, T + 1)
sizes = np.floor(w / w.sum() * E).astype(np.int64)
sizes[0] += E - sizes.sum()
ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
tgt = np.concatenate([np.sort(rng.integers(0, N, s)) for s in sizes]).astype(np.int32)
src = (tgt // 2000 * 2000 + rng.integers(0, 2000, E)).clip(0, N - 1).astype(np.int32)
src, tgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
h, gq = torch.randn(N, Din, device="cuda"), torch.randn(N, Dm, device="cuda")
hp, gqp = ops.pack_bf16x3(h), ops.pack_bf16x3(gq)
bits = torch.randint(-2**31, 2**31 - 1, (E, Dm // 32), device="cuda", dtype=torch.int32) & torch.randint(-2**31, 2**31 - 1, (E, Dm // 32), device="cuda", dtype=torch.int32) \
    & torch.randint(-2**31, 2**31 - 1, (E, Dm // 32), device="cuda", dtype=torch.int32)  # ~12 % set bits
gw = torch.zeros(T, 2 * Din, Dm, device="cuda")
run = lambda: ops.gemm_wgrad_routed_x6([(hp, src, Din), (hp, tgt, Din)], gqp, tgt, bits, E, Dm, gw, gw_group_stride=2 * Din * Dm, group_ptr=ptr, G=T)
for _ in range(3):
    run()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    run()
ev1.record()
torch.cuda.synchronize()
print(f"ablate={ABLATE} trace_build={int(TRACE)} Din={Din} Dm={Dm}: {ev0.elapsed_time(ev1) / 10:.3f} ms per launch")
if not TRACE:
    sys.exit(0)
for wg in (int(a) for a in os.environ.get("WGS", "40,300").split(",")):
    trace = torch.zeros(8 * 64 * 8, dtype=torch.int64, device="cuda")
    assert lib.bl_debug_wgrad_trace(trace.data_ptr(), wg) == 0
    run()
    torch.cuda.synchronize()
    lib.bl_debug_wgrad_trace(None, -1)
    t = trace.cpu().numpy().reshape(8, 64, 8)
    print(f"workgroup {wg}: shader cycles, median over the recorded stages (wave: first 16-message step + staging pieces | second step | barrier wait | whole iteration)")
    for wv in range(8):
        st = t[wv]
        ok = (st[:, 0] > 0) & (st[:, 5] > 0)
        st = st[ok][1:-1]
        if len(st) == 0:
            print(f"  wave {wv}: no stamps")
            continue
        med = lambda x: int(np.median(x))
        print(f"  wave {wv}: {med(st[:, 1] - st[:, 0]):6d} | {med(st[:, 2] - st[:, 1]):6d} | {med(st[:, 5] - st[:, 2]):6d} | iter {med(np.diff(st[:, 0])):6d}  ({len(st)} stages)")

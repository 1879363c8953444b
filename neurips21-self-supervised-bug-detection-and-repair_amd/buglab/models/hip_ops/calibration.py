"""Box calibration and clock / power sampling for bench.py (measurement support; nothing on the training path imports this).

Boxes of the MI355X pool differ by several per cent in the shader clock they hold at the 1 400 W package limit, which is more
than a round's kernel work moves the headline.  `box_calibration()` runs two fixed kernels of the library --
`bl_calib_mfma_bf16` (dense bf16 MFMA from registers) and `bl_calib_stream_copy` (HBM copy) -- and reports what THIS chip
delivers on them; `SmiSampler` reads the shader clock and the package power through librocm_smi64 (ctypes, in-process: a
`rocm-smi` subprocess answers three times a second, this about a thousand times) while the profiled steps run."""
from __future__ import annotations

import ctypes
import statistics
import threading
import time
from typing import Dict, List, Optional

import torch

MFMA_BF16_PAPER_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak (at 2.4 GHz)
HBM_PAPER_TBS = 8.0


def box_calibration(device=None, mfma_ms: float = 300.0, copy_bytes: int = 1 << 31, copy_reps: int = 3) -> Dict[str, float]:
    """-> {"mfma_calib_tflops", "hbm_calib_tbs", ...}.  About 0.5 s of GPU time; call it OUTSIDE any timed region.
    The matrix loop runs for ~mfma_ms so that the package reaches the clock it sustains under load (the first tens of
    milliseconds run at the boost clock); its rate is taken over the second half of the run (two launches, the second timed)."""
    from buglab.models import hip_ops

    lib = hip_ops.load_library()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    stream = torch.cuda.current_stream(dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    wgs = 2 * ncu  # two waves per SIMD: the second one's MFMAs fill the issue gaps of the first
    flop = ctypes.c_double(0.0)

    def mfma(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        hip_ops._check(lib.bl_calib_mfma_bf16(int(iters), int(wgs), sink.data_ptr(), ctypes.byref(flop), stream.cuda_stream), "bl_calib_mfma_bf16")
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1), flop.value

    ms, _ = mfma(20000)  # sizing launch (also the warm-up)
    iters = max(20000, int(20000 * (0.5 * mfma_ms) / max(ms, 1e-3)))
    mfma(iters)  # first half: the package settles
    ms, fl = mfma(iters)
    out = {"mfma_calib_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "mfma_calib_ms": round(ms, 1)}
    # a frozen gathered GEMM of the library: the exact-fp32 MFMA row GEMM (csrc/bl_gemm.hip, unchanged since round 2) on a
    # synthetic problem of the headline layer's shape -- gathers, LDS staging and matrix pipes together draw the power a
    # training step draws, which the register-only loop above does not
    E, N_, Din, Dm, T = 320000, 64000, 128, 128, 16
    g = torch.Generator(device="cpu").manual_seed(0)
    h = torch.randn(N_, Din, generator=g).to(dev)
    W = (torch.randn(T, 2 * Din, Dm, generator=g) / 16.0).to(dev)
    src = torch.randint(0, N_, (E,), generator=g, dtype=torch.int32).to(dev)
    tgt = torch.sort(torch.randint(0, N_, (E,), generator=g, dtype=torch.int32)).values.to(dev)
    ptr = torch.arange(0, E + 1, E // T, dtype=torch.int32).to(dev)
    c = torch.empty((E, Dm), dtype=torch.float32, device=dev)
    run = lambda: hip_ops.gemm_rows([(h, src), (h, tgt)], W, E, Dm, b_group_stride=2 * Din * Dm, ldb=Dm, group_ptr=ptr, G=T, out=c)
    for _ in range(5):
        run()
    reps = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for half in range(2):  # the second half is the measurement (the package has settled)
        e0.record(stream)
        for _ in range(reps):
            run()
        e1.record(stream)
        e1.synchronize()
    gms = e0.elapsed_time(e1) / reps
    out["gemm_calib_tflops"] = round(2.0 * E * 2 * Din * Dm / (gms * 1e-3) / 1e12, 2)
    out["gemm_calib_kernel"] = "bl_gemm_rows (exact fp32 MFMA, gathered, grouped) E=320000 K=256 N=128, back to back"
    del h, W, src, tgt, ptr, c
    n = int(copy_bytes) // 16 * 16
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.zero_()
    best = None
    for rep in range(copy_reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        hip_ops._check(lib.bl_calib_stream_copy(src.data_ptr(), dst.data_ptr(), n, stream.cuda_stream), "bl_calib_stream_copy")
        e1.record(stream)
        e1.synchronize()
        if rep > 0:  # the first pass faults the pages in
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
    out["hbm_calib_tbs"] = round(2.0 * n / (best * 1e-3) / 1e12, 3)
    out["hbm_calib_ms"] = round(best, 3)
    del src, dst
    return out


class _rsmi_frequencies_t(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * 33)]


class SmiSampler:
    """Samples shader clock (MHz) and package power (W) of the busiest visible device from a thread:
        with SmiSampler() as s: run_steps()
        s.summary() -> {"sclk_mhz": median over the samples taken under load, "power_w": ..., "samples": n} or None
    Every failure (library missing, call not supported in the container) yields None -- measurement context only."""

    def __init__(self, period_s: float = 0.002, load_watts: float = 500.0):
        self.period_s, self.load_watts = period_s, load_watts
        self.samples: List[tuple] = []
        self._stop = False
        self._thread: Optional[threading.Thread] = None
        self._lib = None
        self._ndev = 0
        try:
            lib = ctypes.CDLL("librocm_smi64.so")
            lib.rsmi_init.argtypes = [ctypes.c_uint64]
            lib.rsmi_num_monitor_devices.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
            lib.rsmi_dev_gpu_clk_freq_get.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(_rsmi_frequencies_t)]
            lib.rsmi_dev_current_socket_power_get.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
            if lib.rsmi_init(0) == 0:
                n = ctypes.c_uint32(0)
                if lib.rsmi_num_monitor_devices(ctypes.byref(n)) == 0 and n.value > 0:
                    self._lib, self._ndev = lib, int(n.value)
        except Exception:
            self._lib = None

    def _read(self, dv: int):
        f = _rsmi_frequencies_t()
        p = ctypes.c_uint64(0)
        mhz = watts = None
        if self._lib.rsmi_dev_gpu_clk_freq_get(dv, 0, ctypes.byref(f)) == 0 and f.current < 33:  # RSMI_CLK_TYPE_SYS
            mhz = f.frequency[f.current] / 1e6
        if self._lib.rsmi_dev_current_socket_power_get(dv, ctypes.byref(p)) == 0:
            watts = p.value / 1e6
        return mhz, watts

    def _run(self):
        # a box exposes the sensors of all its GPUs: every device is sampled, summary() keeps the one that drew the most power
        while not self._stop:
            now = time.perf_counter()
            for dv in range(self._ndev):
                mhz, watts = self._read(dv)
                self.samples.append((now, dv, mhz, watts))
            time.sleep(self.period_s)

    def __enter__(self):
        if self._lib is not None:
            self._stop = False
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self) -> Optional[Dict[str, float]]:
        per_dev: Dict[int, list] = {}
        for _, dv, m, w in self.samples:
            if m is not None and w is not None:
                per_dev.setdefault(dv, []).append((m, w))
        if not per_dev:
            return None
        dv = max(per_dev, key=lambda d: statistics.mean(w for _, w in per_dev[d]))
        busy = [(m, w) for m, w in per_dev[dv] if w >= self.load_watts]
        if not busy:
            return None
        return {"sclk_mhz": round(statistics.median(m for m, _ in busy), 0), "power_w": round(statistics.median(w for _, w in busy), 0),
                "samples": len(busy), "smi_device": dv,
                "source": "librocm_smi64 (rsmi_dev_gpu_clk_freq_get SYS, rsmi_dev_current_socket_power_get), samples at >= %d W" % self.load_watts}

#!/usr/bin/env python
"""Turn the raw outputs of tools/collect_profiles.sh (gpurun_out/<tag>_*) into the committed summaries under
profiles/:  python tools/summarize_profiles.py r02f"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
G, P = "gpurun_out", "profiles"
fetch = json.load(open(f"{G}/{tag}_pmc_FETCH_SIZE.json"))
write = json.load(open(f"{G}/{tag}_pmc_WRITE_SIZE.json"))
kernels = {}
for k in set(fetch) | set(write):
    kernels[k] = {"FETCH_SIZE_KB_per_launch": fetch.get(k, {}).get("per_launch", 0.0), "WRITE_SIZE_KB_per_launch": write.get(k, {}).get("per_launch", 0.0),
                  "launches": fetch.get(k, write.get(k))["launches"]}
top = dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["FETCH_SIZE_KB_per_launch"] + kv[1]["WRITE_SIZE_KB_per_launch"]) * kv[1]["launches"])[:16])
json.dump({
    "command": "rocprofv3 --pmc FETCH_SIZE (WRITE_SIZE in a separate pass) --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict",
    "note": "averages per dispatch over the run (6 H=128 layers + 2 concat layers per step); KB as reported by rocprofv3; FETCH_SIZE counts a 16 B/lane "
            "streaming read at 1/2 of its bytes on gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled in bench.py's `traffic`; Infinity-Cache hits are "
            "included in both counters",
    "kernels": top}, open(f"{P}/{tag}_bench_hbm_traffic.json", "w"), indent=1)
for f in ("bench.json", "bench_seq.json", "bench_kernel_stats.csv", "bench_serial_kernel_stats.csv", "bench_seq_kernel_stats.csv"):
    if os.path.exists(f"{G}/{tag}_{f}"):
        shutil.copy(f"{G}/{tag}_{f}", f"{P}/{tag}_{f}")
if os.path.exists(f"{G}/{tag}_sq_pmc.json"):  # SQ counter passes of the serial run (tools/pmc_passes.sh): keep the layer kernels
    sq = json.load(open(f"{G}/{tag}_sq_pmc.json"))["kernels"]
    keep = {k: v for k, v in sq.items() if any(s in k for s in ("x6", "h3", "segment_max", "routed_dgrad", "node_bwd", "mp_scatter", "pack_rows"))}
    json.dump({"command": "rocprofv3 --pmc <one group per pass> --kernel-trace -- python bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-predict --no-also",
               "note": "per-dispatch averages; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)",
               "kernels": keep}, open(f"{P}/{tag}_pmc_sq.json", "w"), indent=1, sort_keys=True)
    for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]:
        print(f"{k[:60]:60s} mfma_busy {v.get('mfma_busy_frac')}  l2_hit {v.get('l2_hit_rate')}")
d = json.loads(open(f"{P}/{tag}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: v for k, v in d.items() if k not in ("roofline", "config", "cpu_baseline")})
print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "serial_ms_per_step")})
for name in ("bench_serial", "bench", "bench_seq"):
    path = f"{P}/{tag}_{name}_kernel_stats.csv"
    if not os.path.exists(path):
        continue
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(x["TotalDurationNs"]) for x in rows)
    # steps the process ran (warm-up + timed + bench.py's two profiling passes): one fused clip + Adam launch per step
    steps = next(int(x["Calls"]) for x in rows if x["Name"].startswith("adam_clip_kernel"))
    print(f"--- {name}: kernel ms per step {tot / steps / 1e6:.3f}, {sum(int(x['Calls']) for x in rows) / steps:.0f} launches per step")
    for x in rows[:14]:
        print(f'{x["Name"][:64]:64s} calls/step {int(x["Calls"]) / steps:6.1f}  ms/step {int(x["TotalDurationNs"]) / steps / 1e6:7.3f}  avg us {float(x["AverageNs"]) / 1e3:8.1f}')

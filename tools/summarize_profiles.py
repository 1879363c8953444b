#!/usr/bin/env python
"""Turn the raw rocprofv3 outputs of one measurement round (gpurun_out/<tag>_*) into the committed
summaries under profiles/:  python tools/summarize_profiles.py r01f"""
import collections
import csv
import json
import shutil
import sys

tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(f"gpurun_out/{tag}_pmc_{c}.csv")):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k] += float(r["Counter_Value"])
        n[k] += 1
    for k in acc:
        out.setdefault(k, {})[c + "_KB_per_launch"] = round(acc[k] / n[k], 1)
        out[k]["launches"] = n[k]
top = sorted(out.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE_KB_per_launch", 0) + kv[1].get("WRITE_SIZE_KB_per_launch", 0)) * kv[1]["launches"])[:14]
json.dump({
    "command": "rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE in a separate pass) --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
    "note": "averages per dispatch over the run (6 H=128 layers + 2 concat layers per step); KB as reported by rocprofv3; FETCH_SIZE counts a 16 B/lane "
            "streaming read at 1/2 of its bytes on gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled in bench.py's `traffic`; Infinity-Cache hits are "
            "included in both counters",
    "kernels": dict(top)}, open(f"profiles/{tag}_bench_hbm_traffic.json", "w"), indent=1)
for f in ("bench.json", "bench_kernel_stats.csv", "bench_serial_kernel_stats.csv"):
    shutil.copy(f"gpurun_out/{tag}_{f}", f"profiles/{tag}_{f}")
for k, v in top[:6]:
    print(k, v)
d = json.load(open(f"profiles/{tag}_bench.json"))
r = d["roofline"]
print({k: v for k, v in d.items() if k not in ("roofline", "config")})
print({k: r[k] for k in r if k != "all_gemm_kernels"})
print(r["all_gemm_kernels"])
rows = list(csv.DictReader(open(f"profiles/{tag}_bench_serial_kernel_stats.csv")))
steps = 13
tot = sum(int(x["TotalDurationNs"]) for x in rows)
print("serialized: total kernel ms per step", round(tot / steps / 1e6, 3))
for x in rows[:24]:
    print(f'{x["Name"][:64]:64s} calls/step {int(x["Calls"]) / steps:6.1f}  ms/step {int(x["TotalDurationNs"]) / steps / 1e6:7.3f}  avg us {float(x["AverageNs"]) / 1e3:8.1f}')

#!/usr/bin/env python
"""Per-kernel averages of every counter in one or more rocprofv3 counter_collection.csv files:
    pmc_table.py out.json a_counter_collection.csv [b_counter_collection.csv ...]
-> {kernel name (up to the argument list): {"launches": n, COUNTER: mean value per launch, ...}}.
Derived, when their inputs are present: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 4 SIMDs x CUs)
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE arrives summed over the
8 XCDs -- a 0.254 ms kernel reports 4.98 M = 8 x 0.26 ms x 2.4 GHz -- so the kernel's cycle count is an eighth of it), l2_hit_rate."""
import collections
import csv
import json
import sys

NUM_CUS = 256
NUM_XCDS = 8
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "void ").replace("(anonymous namespace)::", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
out = {}
for k in acc:
    rec = {"launches": max(cnt[k].values())}
    for c in acc[k]:
        rec[c] = round(acc[k][c] / cnt[k][c], 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("GRBM_GUI_ACTIVE"):
        rec["mfma_busy_frac"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / NUM_XCDS * 4 * NUM_CUS), 4)
    if "TCC_HIT_sum" in rec and "TCC_MISS_sum" in rec and rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"] > 0:
        rec["l2_hit_rate"] = round(rec["TCC_HIT_sum"] / (rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"]), 4)
    out[k] = rec
json.dump({"kernels": out}, open(sys.argv[1], "w"), indent=1, sort_keys=True)

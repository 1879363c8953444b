#!/usr/bin/env python
"""Sum rocprofv3 counter_collection.csv per kernel (tuning helper)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(k, r["Counter_Name"])] += 1
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in acc.items():
    if pat in k:
        print(k, {a: (round(b / calls[(k, a)] / 1e6, 3)) for a, b in v.items()}, "(M per launch)")

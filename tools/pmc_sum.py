#!/usr/bin/env python
"""Per-kernel average of one rocprofv3 counter:  pmc_sum.py <counter_collection.csv> <COUNTER>  -> JSON on stdout
{kernel name (up to the argument list): {"per_launch": mean counter value, "launches": n}}."""
import collections
import csv
import json
import sys

counter = sys.argv[2]
acc, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != counter:
        continue
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    acc[k] += float(r["Counter_Value"])
    n[k] += 1
json.dump({k: {"per_launch": round(acc[k] / n[k], 1), "launches": n[k]} for k in acc}, sys.stdout, indent=1)

#!/usr/bin/env python
"""Kernel launches per training step from a rocprofv3 `--kernel-trace --stats` CSV:  launch_census.py <kernel_stats.csv> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
marker = [int(r["Calls"]) for r in rows if "adam_clip" in r["Name"]]
steps = marker[0] if marker else 1
print(f"steps {steps}  launches/step {sum(int(r['Calls']) for r in rows) / steps:.1f}  kernel ms/step {sum(int(r['TotalDurationNs']) for r in rows) / steps / 1e6:.3f}")
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:top]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']) / steps:6.1f}/step  avg {float(r['AverageNs']) / 1e3:7.1f} us")

#!/bin/bash
# GPU box: rocprofv3 kernel stats of the serial bench (exclusive kernel times) -> gpurun_out/TAG_census_kernel_stats.csv
TAG=${1:-census}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_census
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_census -o s -- python $R/bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --no-predict --no-also > $O/${TAG}_census_rocprof.log 2>&1
f=$(find /tmp/prof_census -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/${TAG}_census_kernel_stats.csv
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${TAG}_census_kernel_stats.csv")))
steps = next(int(x["Calls"]) for x in rows if x["Name"].startswith("adam_clip_kernel"))
tot = sum(int(x["TotalDurationNs"]) for x in rows)
print(f"kernel ms per step {tot / steps / 1e6:.3f}, {sum(int(x['Calls']) for x in rows) / steps:.0f} launches per step, {steps} steps")
for x in rows[:45]:
    print(f'{x["Name"][:90]:90s} calls/step {int(x["Calls"]) / steps:6.1f}  ms/step {int(x["TotalDurationNs"]) / steps / 1e6:7.3f}  avg us {float(x["AverageNs"]) / 1e3:8.1f}')
PY

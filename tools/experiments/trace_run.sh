#!/bin/bash
# GPU box: kernel trace (timestamps) of a few serial steps -> gpurun_out/TAG_trace.csv (kernel name, start, end)
TAG=${1:-trace}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o t -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-predict --no-also > $O/${TAG}_trace_rocprof.log 2>&1
f=$(find /tmp/prof_trace -name '*kernel_trace.csv' | head -1)
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full step of the timed region: between the last two adam_clip launches before the profiling passes -> take steps 3..4 by adam index
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_clip_kernel")]
lo, hi = adam[2], adam[3]
out = open("$O/${TAG}_trace.csv", "w")
t0 = int(rows[lo]["End_Timestamp"])
prev_end = t0
for r in rows[lo + 1:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write(f'{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{(s - prev_end) / 1e3:.1f},{r.get("Stream_Id", "")},{r["Kernel_Name"][:70]}\n')
    prev_end = max(prev_end, e)
out.close()
print("step span us", (int(rows[hi]["End_Timestamp"]) - t0) / 1e3, "kernels", hi - lo)
PY

set -u
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
python bench.py --graphs 15 --no-box --no-also --no-cpu-baseline > $O/r06t1_bench_30k.json 2> $O/r06t1_bench_30k.err
python bench.py --graphs 15 --no-box --no-also --no-cpu-baseline --steps 100 --warmup 20 > $O/r06t1_bench_30k_100.json 2>> $O/r06t1_bench_30k.err
cd /tmp
rm -rf /tmp/prof_30k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_30k -o s -- python $R/bench.py --graphs 15 --steps 20 --warmup 5 --no-cpu-baseline --no-predict --no-also --no-box > $O/r06t1_30k_rocprof.log 2>&1
f=$(find /tmp/prof_30k -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/r06t1_bench_30k_kernel_stats.csv
f=$(find /tmp/prof_30k -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python - "$f" > $O/r06t1_30k_trace_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
# last 40 % of the trace = steady-state steps: busy fraction
cut = t0 + int(0.6 * (t1 - t0))
tail = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
busy = 0; last_end = cut
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    s = max(s, last_end)
    if e > s: busy += e - s; last_end = e
print("tail window ms", (t1 - cut) / 1e6, "busy ms", busy / 1e6, "busy frac", busy / (t1 - cut), "kernels", len(tail))
PY

#!/bin/bash
O=gpurun_out; mkdir -p $O
for p in 0 -1 0 -1; do
  python bench.py --no-cpu-baseline --no-also --no-predict --main-stream-priority $p > $O/r04o_bench_m$p.json 2> $O/r04o_bench_m$p.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04o_bench_m$p.json"))
    print("main priority $p:", d["value"], d["ms_per_step"], "serial", d["roofline"]["serial_ms_per_step"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r04o_bench_m$p.err").read()[-800:])
PY
done

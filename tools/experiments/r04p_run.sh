#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/r04p_gputest.log 2>&1; tail -3 $O/r04p_gputest.log
for v in "" "--default-stream" "" "--default-stream"; do
  python bench.py --no-cpu-baseline --no-also --no-predict $v > $O/r04p_bench.json 2> $O/r04p_bench.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04p_bench.json"))
    print("[$v]", d["value"], d["ms_per_step"], "serial", d["roofline"]["serial_ms_per_step"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r04p_bench.err").read()[-800:])
PY
done

#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/r04r_gputest.log 2>&1; tail -3 $O/r04r_gputest.log
bash tools/experiments/trace_run.sh r04r > $O/r04r_trace.log 2>&1; tail -1 $O/r04r_trace.log
python bench.py --no-cpu-baseline --no-also > $O/r04r_bench.json 2> $O/r04r_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04r_bench.json")); print(d["value"], d["ms_per_step"], d["predict_graphs_per_s"], d["roofline"]["serial_ms_per_step"])
rows = [l.rstrip("\n").split(",", 4) for l in open("gpurun_out/r04r_trace.csv")]
i0 = next(i for i, r in enumerate(rows) if r[4].startswith("gather_rows_kernel"))
i1 = next(i for i, r in enumerate(rows) if "node_bwd_kernel" in r[4])
print("heads + loss section us:", float(rows[i1][0]) - float(rows[i0][0]), "kernels", i1 - i0)
for r in rows[i0:i1]: print(r[0], r[1], r[4][:64])
PY

#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/r04zz_gputest.log 2>&1; tail -2 $O/r04z_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04zz_smoke.log 2>&1; tail -1 $O/r04z_smoke.log
python bench.py > $O/r04zz_bench.json 2> $O/r04z_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04zz_bench.json"))
print(d["value"], d["ms_per_step"], d["predict_graphs_per_s"], d["cpu_baseline"]["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("mfma_busy_frac"))
for k, v in d["also"].items(): print(k[:44], v["value"])
PY

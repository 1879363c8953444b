#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -x -q -m gpu -k "not 64" 2>&1 | tail -3
for cfg in "--hidden 256 --graphs 32" "--hidden 256 --graphs 32 --degree powerlaw" ""; do
  python bench.py $cfg --no-cpu-baseline --no-also --steps 20 --warmup 5 > /tmp/b.json 2>/dev/null
  python - "$cfg" <<PY
import json, sys
j=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); k=j["roofline"]["kernels_serial"]
print("[%s]" % sys.argv[1], j["value"], j["ms_per_step"], "segmax", k["segment_max_ln"]["ms_per_step"], "sums", k["node_grad_sums"]["ms_per_step"], "predict", j["predict_graphs_per_s"])
PY
done | tee $O/r05_hubs.log

#!/bin/bash
# A/B: per-node kernels without the (identity) node_order load at c2 -- same box, alternating
O=gpurun_out; mkdir -p $O
for i in 1 2; do
for v in "" 1; do
  BL_AB_KEEP_NODE_ORDER=$v python bench.py --no-cpu-baseline --no-also --no-predict --steps 20 --warmup 5 > /tmp/b.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); k=j["roofline"]["kernels_serial"]
print("keep_order=${v:-0}", j["value"], j["ms_per_step"], "segmax", k["segment_max_ln"]["ms_per_step"], "sums", k["node_grad_sums"]["ms_per_step"], "serial", j["roofline"]["serial_ms_per_step"])
PY
done; done | tee $O/r05o_node_order_ab.log
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -2

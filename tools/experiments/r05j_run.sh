#!/bin/bash
TAG=${1:-r05j}
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_seq_great_gpu.py tests/test_hip_kernels.py -x -q -m gpu > $O/${TAG}_gputest.log 2>&1; tail -3 $O/${TAG}_gputest.log
python bench.py --model seq-great --no-cpu-baseline --no-also > $O/${TAG}_bench_seq.json 2>$O/${TAG}_bench_seq.err; python - <<PY
import json
f="$O/${TAG}_bench_seq.json"
try:
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in list(j["roofline"]["kernels_serial"].items())[:14]})
except Exception as e: print(f, "ERR", e); print(open("$O/${TAG}_bench_seq.err").read()[-2000:])
PY
